#!/bin/bash
# developer aid: rocprofv3 kernel-trace of the whole default bench (all legs), per-kernel stats
R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp; export TMPDIR=/tmp RPL_SYNTH_CACHE=/tmp/rplc
mkdir -p $R/gpurun_out/rk; rm -rf /tmp/rk
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rk -o k -- python $R/bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-single > $R/gpurun_out/rk/run.log 2>&1
f=$(find /tmp/rk -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/rk/kernel_stats.csv
python - <<PY
import csv
for r in csv.DictReader(open("$R/gpurun_out/rk/kernel_stats.csv")):
    if r["Name"].startswith(("void rpl", "rpl::")):
        print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:10.1f} us  min {float(r["MinNs"])/1e3:10.1f}  max {float(r["MaxNs"])/1e3:10.1f}')
PY
