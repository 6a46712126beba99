#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/r4g; mkdir -p $O; export RPL_SYNTH_CACHE=/tmp/rplc
{
RPLGPU_VOXEL_PATH=two RPLGPU_VOXEL_PIPE=1 timeout 120 python tools/dev/pipetl.py 2>&1 | tail -16
for pipe in 1 2; do RPLGPU_VOXEL_PIPE=$pipe timeout 120 python tools/dev/vbench.py 4096 30 2>&1 | tail -1 | sed "s/^/rfirst pipe=$pipe /"; done
RPLGPU_VOXEL_PIPE=1 timeout 120 python tools/dev/vbench.py 4096 10 0.01 2>&1 | tail -1 | sed "s/^/rfirst pipe=1 /"
RPLGPU_VOXEL_PIPE=1 bash tools/dev/kstats.sh pipe1 4096 10
} 2>&1 | tee $O/timeline2.txt
RPLGPU_VOXEL_PATH=two timeout 600 python -m pytest tests -m gpu -x -q -k "not node_patch" > $O/pytest_two.log 2>&1; echo "two rc=$?"; tail -2 $O/pytest_two.log
