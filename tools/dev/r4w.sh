#!/bin/bash
# 1024-thread fused kernel (one workgroup per CU) against the 512-thread build (two per CU, 4096-entry queue) on rings
# that fit the small queue (r0 <= 12 m) and on the bench distribution (r0 <= 30 m: a third of the scans overflow)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/r4w; mkdir -p $O; export RPL_SYNTH_CACHE=/tmp/rplc
LIB=$R/rplidar_ros2_driver_amd/lib
{ for r0 in 12 30; do for v in base t512; do
  L=$LIB/librplgpu_$v.so; [ $v = base ] && L=$LIB/librplgpu.so
  echo "== $v r0max=$r0"; RPL_VOXDBG_R0MAX=$r0 RPLGPU_LIBRARY=$L timeout 120 python tools/voxdbg.py 4096 2>&1 | egrep "kernel ms|records|stream|total" | tail -5 | tr '\n' ' '; echo
done; done; } 2>&1 | tee $O/t512.txt
