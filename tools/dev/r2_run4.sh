#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export RPL_SYNTH_CACHE=/tmp/rplc
OUT=$R/gpurun_out/r2_4; mkdir -p $OUT
L=$R/rplidar_ros2_driver_amd/lib
for i in 1 2; do for v in A AUX0 AUX1 GLOB; do echo -n "$v: "; RPLGPU_LIBRARY=$L/librplgpu_$v.so timeout 120 python tools/voxdbg.py 1024 2>&1 | egrep "kernel ms|stream|total mean" | tail -3 | tr '\n' ' '; echo; done; done | tee $OUT/aux.txt
