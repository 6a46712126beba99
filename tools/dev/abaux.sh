#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export RPL_SYNTH_CACHE=/tmp/rplc
for v in 2 0 1 3 2 0; do
  [ $v = 2 ] && unset RPLGPU_LIBRARY || export RPLGPU_LIBRARY=$R/rplidar_ros2_driver_amd/lib/librplgpu_aux$v.so
  echo "== aux $v"; timeout 120 python tools/voxdbg.py 4096 2>&1 | grep -E "kernel ms|stream|total" | tail -3
done
