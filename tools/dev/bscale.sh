#!/bin/bash
# launch time of the headline kernel against the number of scans (per-item cost = slope, fixed cost + tail = intercept)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/bscale; mkdir -p $O; export RPL_SYNTH_CACHE=/tmp/rplc
for B in ${BS:-256 512 1024 2048 4096 8192 16384}; do timeout 300 python tools/dev/vbench.py $B 20 2>&1 | tail -1 | sed 's/path=auto stage=- //; s/status=0 //'; done | tee $O/bscale.txt
