#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
bash tools/dev/rg.sh "decode or unpack or stream or recorded"
timeout 300 python tools/dev/decbench.py 4096 2>&1 | grep -E "ans|segment"
