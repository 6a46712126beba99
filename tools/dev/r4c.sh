#!/bin/bash
# round 4, GPU call C: the pipelined two-kernel path (k_voxel_cells next to k_voxel_runs)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/r4c; mkdir -p $O; export RPL_SYNTH_CACHE=/tmp/rplc
{
for pipe in 1 0 2; do RPLGPU_VOXEL_PIPE=$pipe timeout 120 python tools/dev/vbench.py 4096 30 2>&1 | tail -1 | sed "s/^/pipe=$pipe /"; done
RPLGPU_VOXEL_PATH=fused timeout 120 python tools/dev/vbench.py 4096 30 2>&1 | tail -1
RPLGPU_VOXEL_PIPE=1 timeout 120 python tools/dev/vbench.py 4096 10 0.01 2>&1 | tail -1 | sed "s/^/pipe=1 /"
RPLGPU_VOXEL_PIPE=1 timeout 120 python tools/dev/vbench.py 512 10 0 uniform 2>&1 | tail -1 | sed "s/^/pipe=1 /"
RPLGPU_VOXEL_PATH=fused timeout 120 python tools/dev/vbench.py 512 10 0 uniform 2>&1 | tail -1
} 2>&1 | tee $O/vbench.txt
RPLGPU_VOXEL_PATH=two timeout 600 python -m pytest tests -m gpu -x -q -k "not node_patch" > $O/pytest_two.log 2>&1; echo "two rc=$?"; tail -4 $O/pytest_two.log
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_auto.log 2>&1; echo "auto rc=$?"; tail -3 $O/pytest_auto.log
RPLGPU_VOXEL_PIPE=1 bash tools/dev/kstats.sh pipe1 4096 10 2>&1 | tee $O/kstats.txt
cp /tmp/ks_pipe1/*/*kernel_trace.csv $O/ 2>/dev/null; cp /tmp/ks_pipe1/*kernel_trace.csv $O/ 2>/dev/null; ls $O
