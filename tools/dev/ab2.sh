#!/bin/bash
# tools/dev/ab2.sh — same-box A/B of lib/librplgpu_A.so vs _B.so on clean (C3-like) and noisy scans
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export RPL_SYNTH_CACHE=/tmp/rplc
for i in 1 2; do for v in ${AB_V:-A B}; do
  for cfg in "2048 0.0" "256 0.01"; do
    echo -n "$v [$cfg]: "; RPLGPU_LIBRARY=$R/rplidar_ros2_driver_amd/lib/librplgpu_$v.so timeout 100 python tools/voxdbg.py $cfg 2>&1 | egrep "kernel ms|stream|rank|total mean" | tail -4 | tr -s ' ' | tr '\n' ' '; echo
  done; done; done
