#!/bin/bash
# round 4, GPU call A: parity of the two-kernel voxel path (forced for every batch size, then auto) + first timings
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/r4a; mkdir -p $O; export RPL_SYNTH_CACHE=/tmp/rplc
RPLGPU_VOXEL_PATH=two timeout 900 python -m pytest tests -m gpu -x -q -k "not node_patch" > $O/pytest_two.log 2>&1; echo "two rc=$?"; tail -5 $O/pytest_two.log
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_auto.log 2>&1; echo "auto rc=$?"; tail -3 $O/pytest_auto.log
for path in fused auto; do RPLGPU_VOXEL_PATH=$path timeout 200 python tools/dev/vbench.py 4096 30 2>&1 | tail -1; done | tee $O/vbench.txt
for st in 256 512 2048; do RPLGPU_VOXEL_STAGE=$st RPLGPU_REGION_MB=1200 timeout 200 python tools/dev/vbench.py 4096 30 2>&1 | tail -1; done | tee -a $O/vbench.txt
RPLGPU_RUNS_SCAN_MAJOR=1 timeout 200 python tools/dev/vbench.py 4096 30 2>&1 | tail -1 | tee -a $O/vbench.txt
timeout 200 python tools/dev/vbench.py 4096 10 0.01 2>&1 | tail -1 | tee -a $O/vbench.txt
RPLGPU_VOXEL_PATH=fused timeout 200 python tools/dev/vbench.py 4096 10 0.01 2>&1 | tail -1 | tee -a $O/vbench.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/tools/dev/vbench.py 4096 10 > $O/stats.log 2>&1
cd $R; python - <<PY
import csv,glob
for f in glob.glob("$O/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:70], r["Calls"], r["AverageNs"], r["MinNs"], r["Percentage"])
PY
