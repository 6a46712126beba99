#!/bin/bash
# developer aid: the LDS-staged decoder against the plain one (RPLGPU_DEC_STAGE=0), optionally in variant libraries
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/decstg; mkdir -p $O
{ for m in ${MODES:-plain staged}; do
  echo "== $m ${RPLGPU_LIBRARY##*/}"; unset RPLGPU_DEC_STAGE
  [ $m = plain ] && export RPLGPU_DEC_STAGE=0
  DEC_SUM=1 timeout 300 python tools/dev/decbench.py 4096 2>&1 | grep -v "amdgpu.ids\|^  segment\|checksum"
done; } 2>&1 | tee -a $O/out.txt
