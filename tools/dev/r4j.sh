#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/r4j; mkdir -p $O; export RPL_SYNTH_CACHE=/tmp/rplc
timeout 1200 python -m pytest tests -m gpu -x -q --durations=6 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
timeout 900 python bench.py --steps 30 --cpu-seconds 4 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<PY
import json
l=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k:l[k] for k in ("value","ms_per_step","n_gpus")}, l["roofline"]["kernel_ms_avg"], l["roofline"]["frac"], l["roofline"]["traffic"], l["roofline"]["traffic_source"])
print(json.dumps(l["reference_path_gpu"], indent=1)[:3000])
print(l["cpu_baseline"]["cpu_model"], l["cpu_baseline"]["cores"])
print({k:(v["ms"],v["frac"]) for k,v in l["variants"].items()}, l["c5"]["ms"], l["c5"]["fused_grid"]["ms"])
PY
