#!/bin/bash
# round 4, GPU call B: what bounds k_voxel_runs (ablations, occupancy, prefetch depth) and k_voxel_cells (stage size)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/r4b; mkdir -p $O; export RPL_SYNTH_CACHE=/tmp/rplc
LIB=$R/rplidar_ros2_driver_amd/lib
{
bash tools/dev/kstats.sh base 4096 10
for v in nostore nogather noraw nosg w4 a2 t512; do RPLGPU_LIBRARY=$LIB/librplgpu_$v.so bash tools/dev/kstats.sh $v 4096 10; done
RPLGPU_VOXEL_STAGE=4096 RPLGPU_REGION_MB=2400 bash tools/dev/kstats.sh stage4096 4096 10
RPLGPU_VOXEL_STAGE=2048 RPLGPU_REGION_MB=2400 bash tools/dev/kstats.sh stage2048 4096 10
} 2>&1 | tee $O/ablate.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $O/counters.txt 2>&1
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM" \
           "TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum FETCH_SIZE WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$i -o p -- python $R/tools/dev/vbench.py 4096 3 > $O/pmc_$i.log 2>&1
  echo "pmc $i rc=$?"
done
cd $R; python - <<PY
import csv,glob,collections
for d in sorted(glob.glob("$O/pmc_*")):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:30]
            if "voxel" in k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in agg:
        print(d.split("/")[-1], k, {c: round(sum(v)/len(v)) for c,v in agg[k].items()})
PY
