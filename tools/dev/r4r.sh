#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/r4r; mkdir -p $O; export RPL_SYNTH_CACHE=/tmp/rplc
LIB=$R/rplidar_ros2_driver_amd/lib
{ for i in 1 2; do for v in base fa3; do
  L=$LIB/librplgpu_$v.so; [ $v = base ] && L=$LIB/librplgpu.so
  RPLGPU_LIBRARY=$L timeout 120 python tools/dev/vbench.py 4096 30 2>&1 | tail -1 | sed "s/^/$v /"
  RPLGPU_LIBRARY=$L timeout 120 python tools/dev/vbench.py 4096 10 0.01 2>&1 | tail -1 | sed "s/^/$v /"
done; done; } 2>&1 | tee $O/ahead3.txt
