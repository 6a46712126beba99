#!/bin/bash
# tools/dev/ab.sh — same-box A/B of two builds (lib/librplgpu_A.so vs _B.so): phase cycles, alternating
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export RPL_SYNTH_CACHE=/tmp/rplc
for i in 1 2 3; do for v in A B C; do echo -n "$v: "; RPLGPU_LIBRARY=$R/rplidar_ros2_driver_amd/lib/librplgpu_$v.so timeout 100 python tools/voxdbg.py ${VX_B:-2048} 2>&1 | egrep "kernel ms|stream|total mean" | tail -3 | tr '\n' ' '; echo; done; done
