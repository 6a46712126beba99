#!/bin/bash
# developer aid: per-kernel times of the decode stage for one answer type (DEC_ONLY=0x85 ...)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/rdp; rm -rf /tmp/rdp
DEC_ONLY=${1:-0x85} timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rdp -o dec -- python tools/dev/decbench.py 4096 > gpurun_out/rdp/run.log 2>&1
f=$(find /tmp/rdp -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/rdp/kernel_stats.csv
cut -d, -f1-8 gpurun_out/rdp/kernel_stats.csv | head -12; grep -E "ans|segment" gpurun_out/rdp/run.log
