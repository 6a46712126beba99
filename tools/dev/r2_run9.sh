#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export RPL_SYNTH_CACHE=/tmp/rplc
L=$R/rplidar_ros2_driver_amd/lib
for i in 1 2; do for v in P1 RING; do echo -n "$v: "; RPLGPU_LIBRARY=$L/librplgpu_$v.so timeout 120 python tools/voxdbg.py 1024 2>&1 | egrep "kernel ms|stream|total mean|status" | tail -4 | tr '\n' ' '; echo; done; done
RPLGPU_LIBRARY=$L/librplgpu_RING.so timeout 900 python -m pytest tests -m gpu -q -x -k "voxel or cloud or c5 or scale or fuzz or ror" 2>&1 | tail -3
