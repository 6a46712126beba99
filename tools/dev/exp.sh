#!/bin/bash
# tools/dev/exp.sh "<exp values>" — phase cycles of k_cloud_voxel under RPLGPU_EXP settings
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export RPL_SYNTH_CACHE=/tmp/rplc
mkdir -p gpurun_out/exp
for e in $1; do echo "=== RPLGPU_EXP=$e"; RPLGPU_EXP=$e timeout 200 python tools/voxdbg.py 1024 2>&1 | grep -v amdgpu.ids | tee gpurun_out/exp/exp_$e.txt | egrep "kernel ms|stream|total|records"; done
