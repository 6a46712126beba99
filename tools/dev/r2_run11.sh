#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export RPL_SYNTH_CACHE=/tmp/rplc
L=$R/rplidar_ros2_driver_amd/lib
export RPL_VOXDBG_R0MAX=12 RPL_VOXDBG_CLK=1
echo -n "FC g256: "; RPLGPU_VOXEL_GRID=256 RPLGPU_LIBRARY=$L/librplgpu_FC.so timeout 120 python tools/voxdbg.py 2048 2>&1 | egrep "kernel ms|stream|total mean|core clock" | tail -4 | tr '\n' ' '; echo
echo -n "FC g8: "; RPLGPU_VOXEL_GRID=8 RPLGPU_LIBRARY=$L/librplgpu_FC.so timeout 120 python tools/voxdbg.py 2048 2>&1 | egrep "kernel ms|stream|total mean|core clock" | tail -4 | tr '\n' ' '; echo
echo -n "HC g256: "; RPLGPU_VOXEL_GRID=256 RPLGPU_LIBRARY=$L/librplgpu_HC.so timeout 120 python tools/voxdbg.py 2048 2>&1 | egrep "kernel ms|stream|total mean|core clock" | tail -4 | tr '\n' ' '; echo
echo -n "HC g512: "; RPLGPU_VOXEL_GRID=512 RPLGPU_LIBRARY=$L/librplgpu_HC.so timeout 120 python tools/voxdbg.py 2048 2>&1 | egrep "kernel ms|stream|total mean|core clock" | tail -4 | tr '\n' ' '; echo
echo -n "NAC g256: "; RPLGPU_VOXEL_GRID=256 RPLGPU_LIBRARY=$L/librplgpu_NAC.so timeout 120 python tools/voxdbg.py 2048 2>&1 | egrep "kernel ms|stream|total mean|core clock" | tail -4 | tr '\n' ' '; echo
