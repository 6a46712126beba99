#!/bin/bash
# tools/dev/tapmc.sh — TA / TCP counters of the voxel kernels (fused path, then the two-kernel path), a few
# counters per pass (more "exceeds the capabilities of the hardware"), every pass under its own timeout.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/tapmc; mkdir -p $O; export RPL_SYNTH_CACHE=/tmp/rplc
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "TCP_TA_ADDR_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  for path in fused two; do
    RPLGPU_VOXEL_PATH=$path timeout 90 rocprofv3 --pmc $grp --output-format csv -d $O/${path}_$i -o p -- python $R/tools/dev/vbench.py 4096 2 > $O/${path}_$i.log 2>&1
    echo "pass $i $path rc=$?"
  done
done
cd $R; python - <<PY
import csv,glob,collections
for d in sorted(glob.glob("$O/*_[0-9]*")):
    if not d.split("/")[-1][0] in "ft": continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "voxel" in k: agg[k.split("(")[0].split("::")[-1][:24]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in agg:
        print(d.split("/")[-1], k, {c: round(sum(v)/len(v)) for c,v in agg[k].items()})
PY
