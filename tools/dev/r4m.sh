#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/r4m; mkdir -p $O; export RPL_SYNTH_CACHE=/tmp/rplc
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
RPL_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 30 --cpu-seconds 0 --no-laserscan --no-variants --no-decode --no-single > $O/bench_dist1.json 2> $O/bench_dist1.err; echo "bench rc=$?"
python - <<PY
import json
l=json.loads(open("$O/bench_dist1.json").read().strip().splitlines()[-1])
print({k:l[k] for k in ("value","ms_per_step","compute_only_ms","exchange_only_ms","gathered_bytes_per_rank","exchange_backend","exchange_ranks","status_bits")})
PY
