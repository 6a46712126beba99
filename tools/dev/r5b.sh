#!/bin/bash
# round 5, call b: full GPU suite + bench line (ascend local repair)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/${1:-r5b}; mkdir -p $O; export RPL_SYNTH_CACHE=/tmp/rplc
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -2 $O/bench.err
python - <<P
import json
d=json.load(open("$O/bench.json"))
print("ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"])
print(json.dumps(d["reference_path_gpu"], indent=0)[:1800])
print({k:(v["ms"],v["frac"]) for k,v in d["variants"].items()})
print({k:(v.get("ms"),v.get("frac")) for k,v in d["decode"].items() if isinstance(v,dict)})
P
