#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/r4f; mkdir -p $O; export RPL_SYNTH_CACHE=/tmp/rplc
LIB=$R/rplidar_ros2_driver_amd/lib
{
for pipe in 1 2; do RPLGPU_VOXEL_PIPE=$pipe timeout 120 python tools/dev/vbench.py 4096 30 2>&1 | tail -1 | sed "s/^/prio3 pipe=$pipe /"; done
for v in p0 p1; do RPLGPU_LIBRARY=$LIB/librplgpu_$v.so RPLGPU_VOXEL_PIPE=1 timeout 120 python tools/dev/vbench.py 4096 30 2>&1 | tail -1 | sed "s/^/$v pipe=1 /"; done
RPLGPU_VOXEL_PIPE=1 timeout 120 python tools/dev/vbench.py 4096 10 0.01 2>&1 | tail -1 | sed "s/^/prio3 pipe=1 /"
RPLGPU_VOXEL_PIPE=1 bash tools/dev/kstats.sh pipe1 4096 10
} 2>&1 | tee $O/vbench.txt
