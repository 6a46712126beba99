#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export RPL_SYNTH_CACHE=/tmp/rplc
L=$R/rplidar_ros2_driver_amd/lib
for v in LAT LATNG LATNR; do for g in 256 8; do echo "== $v grid $g"; RPLGPU_VOXEL_GRID=$g RPLGPU_LIBRARY=$L/librplgpu_$v.so timeout 120 python tools/voxdbg.py 1024 2>&1 | egrep "stream|load\+|rowhist" | tail -3; done; done
