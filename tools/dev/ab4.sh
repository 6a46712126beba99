#!/bin/bash
# tools/dev/ab4.sh — same-box A/B of lib/librplgpu_base.so against lib/librplgpu.so: phase cycles and launch times
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/ab4; mkdir -p $O; export RPL_SYNTH_CACHE=/tmp/rplc
LIB=$R/rplidar_ros2_driver_amd/lib
{ for i in 1 2; do for v in base new; do
  L=$LIB/librplgpu.so; [ $v = base ] && L=$LIB/librplgpu_base.so
  echo "== $v"; RPLGPU_LIBRARY=$L timeout 120 python tools/voxdbg.py 2048 2>&1 | egrep "rank|emit|total" | tr '\n' ' '; echo
  RPLGPU_LIBRARY=$L timeout 120 python tools/dev/vbench.py 4096 30 2>&1 | tail -1
  RPLGPU_LIBRARY=$L timeout 120 python tools/dev/vbench.py 4096 10 0.01 2>&1 | tail -1
done; done; } 2>&1 | tee $O/ab.txt
timeout 600 python -m pytest tests/test_gpu_scale.py tests/test_gpu_fuzz.py -m gpu -x -q -k "not decode" 2>&1 | tail -2
