#!/bin/bash
# tools/dev/c5prof.sh — rocprofv3 of config 5 (tools/dev/c5bench.py): kernel trace of both E5 modes, then HBM traffic
# (FETCH_SIZE / WRITE_SIZE, separate counter-only passes) of each mode.  Output: gpurun_out/c5prof/
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/c5prof; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; export RPL_SYNTH_CACHE=/tmp/rplc; export C5_ONLY=${C5_ONLY:-arena}; export VB_AGG=${VB_AGG:-2}
CMD="python $R/tools/dev/c5bench.py 4096 3"
C5_ROUNDS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $CMD > $OUT/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  C5_ROUNDS=1 timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- $CMD > $OUT/pmc_$c.log 2>&1
done
cd $R
python - <<PY
import csv, glob, collections
out = "$OUT"
f = glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True)
print("== kernel stats (tools/dev/c5bench.py 4096 3, C5_ONLY=$C5_ONLY, two-class aggregation pinned: E5 inside (k_cloud_voxel<..., 1>), then two kernels (k_ror_mask + k_cloud_voxel<..., 0>))")
for row in csv.DictReader(open(f[0])):
    if float(row["Percentage"]) > 0.5:
        print("  %-60s calls %4s avg %9.1f us min %9.1f us" % (row["Name"][:60], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(out + "/pmc_%s/**/*counter_collection.csv" % c, recursive=True)
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        if row["Counter_Name"] == c:
            agg[row["Kernel_Name"][:70]].append(float(row["Counter_Value"]))
    print("==", c, "(KiB per dispatch, mean over dispatches; FETCH_SIZE x 2 on gfx950)")
    for k, v in agg.items():
        if sum(v) / len(v) > 1000:
            print("  %-70s n %3d mean %12.1f" % (k, len(v), sum(v) / len(v)))
PY
