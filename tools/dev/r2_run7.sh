#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export RPL_SYNTH_CACHE=/tmp/rplc
L=$R/rplidar_ros2_driver_amd/lib
for i in 1 2; do for v in P1 P2 P3 P4; do echo -n "$v: "; RPLGPU_LIBRARY=$L/librplgpu_$v.so timeout 120 python tools/voxdbg.py 1024 2>&1 | egrep "kernel ms|stream|total mean|status" | tail -4 | tr '\n' ' '; echo; done; done
echo "== P2W"; RPLGPU_LIBRARY=$L/librplgpu_P2W.so timeout 120 python tools/voxdbg.py 1024 2>&1 | egrep "stream|load\+|rowhist" | tail -3
RPLGPU_LIBRARY=$L/librplgpu_P2.so timeout 600 python -m pytest tests -m gpu -q -x -k "voxel or cloud or c5 or scale" 2>&1 | tail -3
