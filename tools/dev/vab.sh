#!/bin/bash
# tools/dev/vab.sh [tags...] — same-box A/B of the shipped library against lib/librplgpu_<tag>.so builds (tools/dev/mkv.sh
# or a copy of an earlier build): tools/dev/vbench.py on the clean C3 batch, alternating, three passes; then 1 cm noise once
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/vab; mkdir -p $O; export RPL_SYNTH_CACHE=/tmp/rplc
LIB=$R/rplidar_ros2_driver_amd/lib
run() { local v=$1; shift; local L=$LIB/librplgpu_$v.so; [ $v = new ] && L=$LIB/librplgpu.so
  env "$@" RPLGPU_LIBRARY=$L timeout 120 python tools/dev/vbench.py ${VB_B:-4096} 30 ${VB_NOISE:-0} ${VB_KIND:-ring} 2>&1 | tail -1 | sed 's/status=0 //'; }
{ for rep in 1 2 3; do for v in new "$@"; do echo -n "[clean $v] "; run $v; done; done
  for v in new "$@"; do echo -n "[noise1cm $v] "; VB_NOISE=0.01 run $v; done
  for v in new "$@"; do echo -n "[r0<=12 $v] "; run $v VB_R0MAX=12; done; } 2>&1 | tee $O/vab.txt
