#!/bin/bash
# tools/dev/abdec.sh — same-box A/B of lib/librplgpu_A.so vs _B.so on the decode stage
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for i in 1 2; do for v in ${AB_V:-A B}; do
  echo "== $v"; RPLGPU_LIBRARY=$R/rplidar_ros2_driver_amd/lib/librplgpu_$v.so timeout 150 python tools/dev/decbench.py ${1:-4096} 2>&1 | grep "^ans\|segment" | cut -c1-60
done; done
