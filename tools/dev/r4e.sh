#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/r4e; mkdir -p $O; export RPL_SYNTH_CACHE=/tmp/rplc
LIB=$R/rplidar_ros2_driver_amd/lib
{
for pipe in 1 0 2; do RPLGPU_VOXEL_PIPE=$pipe timeout 120 python tools/dev/vbench.py 4096 30 2>&1 | tail -1 | sed "s/^/pipe=$pipe /"; done
RPLGPU_LIBRARY=$LIB/librplgpu_c6.so RPLGPU_VOXEL_PIPE=1 timeout 120 python tools/dev/vbench.py 4096 30 2>&1 | tail -1 | sed "s/^/c6 pipe=1 /"
RPLGPU_VOXEL_PIPE=1 timeout 120 python tools/dev/vbench.py 4096 10 0.01 2>&1 | tail -1 | sed "s/^/pipe=1 /"
RPLGPU_VOXEL_PIPE=1 bash tools/dev/kstats.sh pipe1 4096 10
RPLGPU_VOXEL_PIPE=0 bash tools/dev/kstats.sh pipe0 4096 10
} 2>&1 | tee $O/vbench.txt
cp /tmp/ks_pipe1/*kernel_trace.csv $O/ 2>/dev/null
RPLGPU_VOXEL_PATH=two timeout 600 python -m pytest tests -m gpu -x -q -k "not node_patch" > $O/pytest_two.log 2>&1; echo "two rc=$?"; tail -2 $O/pytest_two.log
