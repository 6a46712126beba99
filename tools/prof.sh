#!/bin/bash
# tools/prof.sh <tag> [bench args...] — rocprofv3 kernel-trace + PMC passes of bench.py on the
# GPU box.  Counters are collected in their own runs (never combined with tracing domains).
# Output: gpurun_out/prof_<tag>/{stats,pmc_*}/... plus a text summary gpurun_out/prof_<tag>/summary.txt
set -u
TAG=${1:-x}; shift || true
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; export RPL_SYNTH_CACHE=/tmp/rplc
# --no-c5: every k_cloud_voxel dispatch of the profiled command is the headline launch
BENCH="python $R/bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-variants --no-single --no-live-traffic $*"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $BENCH > $OUT/stats.log 2>&1
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$i -o p -- $BENCH > $OUT/pmc_$i.log 2>&1
done
cd $R
python tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
