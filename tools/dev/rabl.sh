#!/bin/bash
# developer aid: phase cycles of several builds of the library (tools/voxdbg.py, clean C3 batch)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export RPL_SYNTH_CACHE=/tmp/rplc
for v in $1; do
  echo "== $v"
  RPLGPU_LIBRARY=$R/rplidar_ros2_driver_amd/lib/librplgpu_$v.so timeout 200 python tools/voxdbg.py ${VX_B:-1024} 2>&1 | grep -E "kernel ms|stream|total mean|records|wave 0" | tail -5
done
