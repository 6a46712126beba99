#!/bin/bash
# A/B of two builds: phase cycles (tools/voxdbg.py) and launch time (tools/dev/vbench.py), alternating
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/r4n; mkdir -p $O; export RPL_SYNTH_CACHE=/tmp/rplc
LIB=$R/rplidar_ros2_driver_amd/lib
{
for i in 1 2; do for v in base new; do
  L=$LIB/librplgpu.so; [ $v = base ] && L=$LIB/librplgpu_base.so
  echo "== $v"; RPLGPU_LIBRARY=$L timeout 120 python tools/voxdbg.py 2048 2>&1 | egrep "kernel ms|stream|load|rowscan|scatter|rank|heads|emit|total" | tail -9
  RPLGPU_LIBRARY=$L timeout 120 python tools/dev/vbench.py 4096 30 2>&1 | tail -1
  RPLGPU_LIBRARY=$L timeout 120 python tools/dev/vbench.py 4096 10 0.01 2>&1 | tail -1
done; done
} 2>&1 | tee $O/ab.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
