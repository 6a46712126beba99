#!/bin/bash
# tools/dev/rg.sh <pytest -k expr> — developer aid: one gpurun call running a slice of the GPU suite
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out/rg
timeout 1200 python -m pytest tests -m gpu -q -x -k "$1" > gpurun_out/rg/pytest.log 2>&1
grep -E "passed|failed|^E " gpurun_out/rg/pytest.log | head -30
