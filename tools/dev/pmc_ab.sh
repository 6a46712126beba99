#!/bin/bash
# tools/dev/pmc_ab.sh "<variants>" — developer aid: rocprofv3 PMC passes (counters only, no tracing
# domains) of tools/voxdbg.py for several builds of librplgpu; prints per-dispatch averages of
# k_cloud_voxel side by side.
R=${GRAFT_REPO_ROOT:-$PWD}; L=$R/rplidar_ros2_driver_amd/lib
OUT=$R/gpurun_out/pmc_ab; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; export RPL_SYNTH_CACHE=/tmp/rplc
PGROUPS=(
 "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"
 "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_IFETCH"
 "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
 "TCP_PERF_SEL_TOTAL_READ TCP_PERF_SEL_TOTAL_HIT_LRU_READ TCP_PERF_SEL_TOTAL_MISS_LRU_READ TCP_PERF_SEL_TOTAL_MISS_EVICT_READ"
 "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH"
)
for v in $1; do
  i=0
  for grp in "${PGROUPS[@]}"; do
    i=$((i+1))
    PYTHONPATH=$R RPLGPU_LIBRARY=$L/librplgpu_$v.so timeout 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/$v/pmc_$i -o p -- python $R/tools/voxdbg.py ${VX_B:-1024} > $OUT/$v.pmc_$i.log 2>&1
  done
done
python - "$OUT" $1 <<'PY'
import csv, glob, sys
from collections import defaultdict
out, variants = sys.argv[1], sys.argv[2:]
tab = defaultdict(dict)
for v in variants:
    agg = defaultdict(list)
    for f in glob.glob(f"{out}/{v}/pmc_*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_cloud_voxel" in row["Kernel_Name"]:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for c, vals in agg.items():
        tab[c][v] = sum(vals) / len(vals)
print("%-40s" % "counter (avg per k_cloud_voxel dispatch)" + "".join("%16s" % v for v in variants))
for c in sorted(tab):
    print("%-40s" % c + "".join("%16.0f" % tab[c].get(v, float("nan")) for v in variants))
PY
