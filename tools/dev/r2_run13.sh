#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export RPL_SYNTH_CACHE=/tmp/rplc
L=$R/rplidar_ros2_driver_amd/lib
export RPL_VOXDBG_R0MAX=9 RPL_VOXDBG_CLK=1
for i in 1 2; do
echo "FC g256: "; RPLGPU_VOXEL_GRID=256 RPLGPU_LIBRARY=$L/librplgpu_FC.so timeout 120 python tools/voxdbg.py 4096 2>&1 | egrep "kernel ms|records|start us" | tail -3
echo "HC g512: "; RPLGPU_VOXEL_GRID=512 RPLGPU_LIBRARY=$L/librplgpu_HC.so timeout 120 python tools/voxdbg.py 4096 2>&1 | egrep "kernel ms|records|start us" | tail -3
done
