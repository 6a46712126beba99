#!/bin/bash
# select-pass depth of the band path (RPL_VOXEL_SEL_NJ): same-box timing of 4 (tree) against variants built with
# tools/dev/mkv.sh nj6 -DRPL_VOXEL_SEL_NJ=6 etc.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/${1:-nj}; mkdir -p $O; export RPL_SYNTH_CACHE=/tmp/rplc
LIB=$R/rplidar_ros2_driver_amd/lib
run() { local v=$1; shift; local L=$LIB/librplgpu_$v.so; [ $v = new ] && L=$LIB/librplgpu.so
  env "$@" RPLGPU_LIBRARY=$L timeout 200 python tools/dev/vbench.py ${VB_B:-4096} ${VB_REPS:-10} ${VB_NOISE:-0} ${VB_KIND:-ring} 2>&1 | tail -1 | sed 's/path=auto stage=- //; s/status=0 //'; }
V="new ${NJ_VARIANTS:-nj6 nj8}"
{ for v in $V $V; do echo -n "[uniform $v] "; VB_KIND=uniform VB_REPS=3 run $v; done
  for v in $V; do echo -n "[noise3cm $v] "; VB_NOISE=0.03 run $v; done
  for v in $V; do echo -n "[noise1cm $v] "; VB_NOISE=0.01 run $v; done; } 2>&1 | tee $O/timing.txt
