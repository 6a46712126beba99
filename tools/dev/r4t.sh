#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export RPL_SYNTH_CACHE=/tmp/rplc
RPL_FUZZ_SEEDS=300 timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -n 8 -k decode 2>&1 | tail -25
