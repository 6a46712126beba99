#!/bin/bash
# tools/dev/vcheck.sh [tags...] — one call for a change of the voxel kernel: the voxel GPU tests (scale + parity, all
# forms), the same-box A/B against lib/librplgpu_<tag>.so builds (vab.sh) and the phase-cycle breakdown (voxdbg.py)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export RPL_SYNTH_CACHE=/tmp/rplc
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 900 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py -m gpu -x -q -k "not bench_batches" 2>&1 | tail -12
fi
bash tools/dev/vab.sh "$@" 2>&1 | grep -v amdgpu.ids
for r in 30 12; do echo "== voxdbg r0max=$r"; RPL_VOXDBG_R0MAX=$r timeout 200 python tools/voxdbg.py 2048 2>&1 | grep -v "amdgpu.ids\|kernel ms\|fast_div"; done
