#!/bin/bash
# developer aid: bench headline + variants for several builds of the library on ONE box
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export RPL_SYNTH_CACHE=/tmp/rplc
for v in $1; do
  RPLGPU_LIBRARY=$R/rplidar_ros2_driver_amd/lib/librplgpu_$v.so python bench.py --cpu-seconds 0 --no-laserscan --no-decode --no-single --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', d['roofline']['kernel_ms_min'], d['roofline']['kernel_ms_avg'], {k:v['ms'] for k,v in d['variants'].items()}, d['c5']['ms'], d['c5']['fused_grid']['ms'])"
done
