#!/bin/bash
# developer aid: A/B of two builds of the library on one box (A = lib/librplgpu_A.so, B = the in-tree build)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export RPL_SYNTH_CACHE=/tmp/rplc
for rep in 1 2; do
for v in A B; do
  [ $v = A ] && export RPLGPU_LIBRARY=$R/rplidar_ros2_driver_amd/lib/librplgpu_A.so || unset RPLGPU_LIBRARY
  echo "== $v"; timeout 120 python tools/voxdbg.py 4096 2>&1 | grep -E "kernel ms|total"
  timeout 200 python bench.py --cpu-seconds 0 --no-laserscan --no-decode --no-single --no-c5 --no-variants --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench', d['ms_per_step'], d['roofline']['kernel_ms_min'], d['roofline']['kernel_ms_avg'])"
done; done
