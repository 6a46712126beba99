#!/bin/bash
# developer aid: kernel durations inside the single-scan calls (what is left of a call is launch + completion latency)
R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp; export TMPDIR=/tmp
mkdir -p $R/gpurun_out/floor
for n in 360 8192; do
  rm -rf /tmp/fl$n
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fl$n -o k -- python $R/tools/dev/single.py $n > $R/gpurun_out/floor/run$n.log 2>&1
  f=$(find /tmp/fl$n -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/floor/kernel_stats_$n.csv
  echo "== n = $n"; grep "us per call" $R/gpurun_out/floor/run$n.log
  python - <<PY
import csv
for r in csv.DictReader(open("$R/gpurun_out/floor/kernel_stats_$n.csv")):
    if "rpl::" in r["Name"]:
        print(f'  {r["Name"].replace("void ","")[:50]:50s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:7.2f} us')
PY
done
