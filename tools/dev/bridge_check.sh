#!/bin/bash
# FILL passes bridging wholly dropped lanes (RPL_VOXEL_BRIDGE = lanes a run may cross): parity at bench scale in all
# forms, then same-box timing of 1 (tree) against 0 / 2 / 3 (tools/dev/mkv.sh br0 -DRPL_VOXEL_BRIDGE=0 ...)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/${1:-bridge}; mkdir -p $O; export RPL_SYNTH_CACHE=/tmp/rplc
timeout 900 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py tests/test_gpu_msg.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2
LIB=$R/rplidar_ros2_driver_amd/lib
run() { local v=$1; shift; local L=$LIB/librplgpu_$v.so; [ $v = new ] && L=$LIB/librplgpu.so
  env "$@" RPLGPU_LIBRARY=$L timeout 200 python tools/dev/vbench.py ${VB_B:-4096} ${VB_REPS:-20} ${VB_NOISE:-0} ${VB_KIND:-ring} 2>&1 | tail -1 | sed 's/path=auto stage=- //; s/status=0 //'; }
V="new ${BR_VARIANTS:-br0 br2 br3}"
{ for rep in 1 2; do for v in $V; do echo -n "[noise1cm $v] "; VB_NOISE=0.01 run $v; done; done
  for v in $V; do echo -n "[noise3cm $v] "; VB_NOISE=0.03 run $v; done
  for v in $V; do echo -n "[clean $v] "; run $v; done; } 2>&1 | tee $O/timing.txt
