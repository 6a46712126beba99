#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export RPL_SYNTH_CACHE=/tmp/rplc
L=$R/rplidar_ros2_driver_amd/lib
export RPL_VOXDBG_R0MAX=12
for i in 1 2; do
echo -n "F g256: "; RPLGPU_VOXEL_GRID=256 RPLGPU_LIBRARY=$L/librplgpu_F.so timeout 120 python tools/voxdbg.py 2048 2>&1 | egrep "kernel ms|records|stream|total mean|status" | tail -5 | tr '\n' ' '; echo
echo -n "H g256: "; RPLGPU_VOXEL_GRID=256 RPLGPU_LIBRARY=$L/librplgpu_H.so timeout 120 python tools/voxdbg.py 2048 2>&1 | egrep "kernel ms|records|stream|total mean|status" | tail -5 | tr '\n' ' '; echo
echo -n "H g512: "; RPLGPU_VOXEL_GRID=512 RPLGPU_LIBRARY=$L/librplgpu_H.so timeout 120 python tools/voxdbg.py 2048 2>&1 | egrep "kernel ms|records|stream|total mean|status" | tail -5 | tr '\n' ' '; echo
done
