#!/bin/bash
# developer aid: voxel parity slice + phase cycles (clean, noisy) + the bench's headline numbers
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export RPL_SYNTH_CACHE=/tmp/rplc
bash tools/dev/rg.sh "voxel or cloud or c5 or scale or fuzz"
timeout 120 python tools/voxdbg.py 4096 2>&1 | tail -12
echo "== noisy 1024"; timeout 120 python tools/voxdbg.py 1024 0.01 2>&1 | tail -12
timeout 300 python bench.py --cpu-seconds 0 --no-laserscan --no-decode --no-single --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench', d['ms_per_step'], d['roofline']['kernel_ms_min'], d['roofline']['kernel_ms_avg'], d['roofline']['frac'])
print({k:(v[\"ms\"],v[\"frac\"]) for k,v in d[\"variants\"].items()}, d[\"c5\"][\"ms\"], d[\"c5\"][\"fused_grid\"][\"ms\"])"
