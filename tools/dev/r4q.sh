#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/r4q; mkdir -p $O; export RPL_SYNTH_CACHE=/tmp/rplc
LIB=$R/rplidar_ros2_driver_amd/lib
{ for i in 1 2; do for v in base og og16 ng; do
  L=$LIB/librplgpu_$v.so; [ $v = base ] && L=$LIB/librplgpu.so
  RPLGPU_LIBRARY=$L timeout 120 python tools/dev/vbench.py 4096 30 2>&1 | tail -1 | sed "s/^/$v /"
done; done; } 2>&1 | tee $O/onegather.txt
