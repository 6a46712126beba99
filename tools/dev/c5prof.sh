#!/bin/bash
# Developer aid: rocprofv3 kernel trace of the config-5 shaped run (tools/dev/rorbench.py).
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/c5prof
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python $R/tools/dev/rorbench.py ${1:-256} > $OUT/log.txt 2>&1
grep -v amdgpu $OUT/log.txt | tail -4
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:8.3f}")
PY
