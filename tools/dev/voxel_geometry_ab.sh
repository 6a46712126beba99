#!/bin/bash
# geometry variants of the fused voxel kernel on one box (tools/dev/vbench.py per library and data set)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/voxel_geometry_ab; mkdir -p $O; export RPL_SYNTH_CACHE=/tmp/rplc
LIB=$R/rplidar_ros2_driver_amd/lib
run() { # tag env...
  local v=$1; shift
  local L=$LIB/librplgpu_$v.so; [ $v = base ] && L=$LIB/librplgpu.so
  env "$@" RPLGPU_LIBRARY=$L timeout 120 python tools/dev/vbench.py ${VB_B:-4096} 30 ${VB_NOISE:-0} ${VB_KIND:-ring} 2>&1 | tail -1
}
{
for rep in 1 2; do
  for v in base d2 t512 a3 d2a3 prio slp6; do echo -n "[clean30 $v] "; run $v; done
done
for v in base d2 t512 d2a3; do echo -n "[clean12 $v] "; run $v VB_R0MAX=12; done
for v in base d2 t512 d2a3; do echo -n "[noise1cm $v] "; VB_NOISE=0.01 run $v; done
for v in base d2 t512; do echo -n "[uniform512 $v] "; VB_B=512 VB_KIND=uniform run $v; done
} 2>&1 | tee $O/variants.txt
echo "== parity of d2 (bench-scale tests)"
RPLGPU_LIBRARY=$LIB/librplgpu_d2.so timeout 600 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_d2.txt
