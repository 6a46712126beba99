#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export RPL_SYNTH_CACHE=/tmp/rplc
OUT=$R/gpurun_out/r2_2; mkdir -p $OUT
L=$R/rplidar_ros2_driver_amd/lib
for i in 1 2; do for v in A NA NT NS NG NR NGR NGRA; do echo -n "$v: "; RPLGPU_LIBRARY=$L/librplgpu_$v.so timeout 120 python tools/voxdbg.py 1024 2>&1 | egrep "kernel ms|stream|total mean" | tail -3 | tr '\n' ' '; echo; done; done | tee $OUT/abl.txt
timeout 1200 python -m pytest tests -m gpu -q -x -k "scale or faults or full_size or large or tie or golden or laserscan_to_cloud" > $OUT/pytest_new.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_new.log
