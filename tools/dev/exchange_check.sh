#!/bin/bash
# exchange tests (gather to root) + forced one-rank distributed bench, both exchange modes
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/${1:-exchange_check}; mkdir -p $O; export RPL_SYNTH_CACHE=/tmp/rplc
timeout 600 python -m pytest tests/test_gpu_comm.py tests/test_gpu2_rccl.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
for x in allgather gather; do
  RPL_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 20 --no-variants --no-decode --no-single --no-laserscan --cpu-seconds 0 --exchange $x > $O/bench_$x.json 2> $O/bench_$x.err; echo "bench $x rc=$?"
  python - <<P
import json
d=json.load(open("$O/bench_$x.json"))
print("$x", "ms_per_step", d["ms_per_step"], {k:d.get(k) for k in ("compute_only_ms","exchange_only_ms","exchange_backend","exchange_ranks")})
print(d.get("compute_only"))
P
done
