#!/bin/bash
# tools/dev/c5check.sh — E5 inside the voxel kernel: its tests, the other tests that run E5 + E4, the C5 timings
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export RPL_SYNTH_CACHE=/tmp/rplc
timeout 1500 python -m pytest tests/test_gpu_ror_inside.py -m gpu -x -q 2>&1 | tail -25
if [ "${SKIP_MORE:-0}" != "1" ]; then
  timeout 1500 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py tests/test_gpu_msg.py tests/test_gpu_comm.py tests/test_gpu_fuzz.py -m gpu -x -q -k "ror or c5 or C5 or fuzz or dropped_lanes or fused or e8 or group" 2>&1 | tail -12
fi
timeout 600 python tools/dev/c5bench.py ${C5_B:-4096} 2>&1 | grep -v amdgpu.ids
