#!/bin/bash
# histogram band cutting of overflow scans (round 5): bench-scale parity in all forms + same-box timing against a
# build without it (tools/dev/mkv.sh nohist -DRPL_VOXEL_NO_HIST_BANDS)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/${1:-bands}; mkdir -p $O; export RPL_SYNTH_CACHE=/tmp/rplc
timeout 900 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py tests/test_gpu_msg.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
LIB=$R/rplidar_ros2_driver_amd/lib
run() { local v=$1; shift; local L=$LIB/librplgpu_$v.so; [ $v = new ] && L=$LIB/librplgpu.so
  env "$@" RPLGPU_LIBRARY=$L timeout 200 python tools/dev/vbench.py ${VB_B:-4096} ${VB_REPS:-10} ${VB_NOISE:-0} ${VB_KIND:-ring} 2>&1 | tail -1 | sed 's/path=auto stage=- //; s/status=0 //'; }
{ for v in new nohist new nohist; do echo -n "[uniform $v] "; VB_KIND=uniform VB_REPS=3 run $v; done
  for v in new nohist; do echo -n "[noise1cm $v] "; VB_NOISE=0.01 run $v; done
  for v in new nohist; do echo -n "[noise3cm $v] "; VB_NOISE=0.03 run $v; done
  for v in new nohist; do echo -n "[clean $v] "; run $v; done; } 2>&1 | tee $O/timing.txt
