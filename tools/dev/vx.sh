#!/bin/bash
# tools/dev/vx.sh <tag> — developer loop for k_cloud_voxel on the GPU box: voxel-related parity
# tests + the phase cycle breakdown (tools/voxdbg.py).  Output in gpurun_out/<tag>/.
set -u
TAG=${1:-vx}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export RPL_SYNTH_CACHE=/tmp/rplc
cd $R
timeout 600 python -m pytest tests -m gpu -x -q -k "${VX_K:-voxel or batch or c5 or full_size or ror}" > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $OUT/pytest.log
timeout 300 python tools/voxdbg.py ${VX_B:-2048} > $OUT/voxdbg.txt 2>&1; cat $OUT/voxdbg.txt
