#!/bin/bash
# tools/dev/kstats.sh <label> [vbench args] — rocprofv3 kernel-trace of tools/dev/vbench.py under the current
# environment; prints avg / min ns of the voxel kernels.
R=${GRAFT_REPO_ROOT:-$PWD}; L=$1; shift; D=/tmp/ks_$L; rm -rf $D; mkdir -p $D
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o s -- python $R/tools/dev/vbench.py "$@" > $D/log.txt 2>&1
tail -1 $D/log.txt | sed "s/^/[$L] /"
python - <<PY
import csv,glob
for f in glob.glob("$D/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "voxel" in r["Name"]:
            print("[$L]   %-46s calls %4s avg %9.1f us min %9.1f us" % (r["Name"][:46], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
