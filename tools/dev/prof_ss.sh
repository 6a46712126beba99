cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ss -o s -- python $R/tools/dev/ssbench.py > $R/gpurun_out/prof_ss.log 2>&1
F=$(find $R/gpurun_out/prof_ss -name "*kernel_stats.csv" | head -1)
python - "$F" <<PY
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"].replace("rpl::","").split("(")[0][:60]
    print("%-62s calls=%-6s avg_us=%8.2f min_us=%8.2f" % (n, r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
