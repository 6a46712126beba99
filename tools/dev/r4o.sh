#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/r4o; mkdir -p $O; export RPL_SYNTH_CACHE=/tmp/rplc
timeout 600 python -m pytest tests/test_gpu_decode.py -m gpu -x -q 2>&1 | tail -30
