#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export RPL_SYNTH_CACHE=/tmp/rplc
OUT=$R/gpurun_out/r2_3; mkdir -p $OUT
L=$R/rplidar_ros2_driver_amd/lib
for v in A B; do for g in 256 128 64 32 8; do echo -n "$v grid=$g: "; RPLGPU_VOXEL_GRID=$g RPLGPU_LIBRARY=$L/librplgpu_$v.so timeout 120 python tools/voxdbg.py 1024 2>&1 | egrep "kernel ms|stream|total mean" | tail -3 | tr '\n' ' '; echo; done; done | tee $OUT/grid.txt
