#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/r4i; mkdir -p $O; export RPL_SYNTH_CACHE=/tmp/rplc
LIB=$R/rplidar_ros2_driver_amd/lib
timeout 1200 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -14 $O/pytest.log
{
timeout 120 python tools/dev/vbench.py 4096 30 2>&1 | tail -1 | sed "s/^/fused /"
RPLGPU_LIBRARY=$LIB/librplgpu_fa3.so timeout 120 python tools/dev/vbench.py 4096 30 2>&1 | tail -1 | sed "s/^/fused-ahead3 /"
timeout 120 python tools/dev/vbench.py 4096 30 2>&1 | tail -1 | sed "s/^/fused /"
} 2>&1 | tee $O/vbench.txt
