#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export RPL_SYNTH_CACHE=/tmp/rplc
L=$R/rplidar_ros2_driver_amd/lib
export RPL_VOXDBG_R0MAX=12 RPL_VOXDBG_CLK=1
echo "FC g256: "; RPLGPU_VOXEL_GRID=256 RPLGPU_LIBRARY=$L/librplgpu_FC.so timeout 120 python tools/voxdbg.py 2048 2>&1 | egrep "kernel ms|core clock|start us|distinct" | tail -4
echo "HC g512: "; RPLGPU_VOXEL_GRID=512 RPLGPU_LIBRARY=$L/librplgpu_HC.so timeout 120 python tools/voxdbg.py 2048 2>&1 | egrep "kernel ms|core clock|start us|distinct" | tail -4
