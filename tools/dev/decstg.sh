#!/bin/bash
# developer aid: the LDS-staged decoder (RPLGPU_DEC_STAGE=0: plain kernel; RPLGPU_DEC_WINDOWS=n: n staging windows)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/decstg; mkdir -p $O
{ for m in ${MODES:-plain auto 1 2 3 4}; do
  echo "== $m"; unset RPLGPU_DEC_STAGE RPLGPU_DEC_WINDOWS
  case $m in plain) export RPLGPU_DEC_STAGE=0;; auto) ;; *) export RPLGPU_DEC_WINDOWS=$m;; esac
  DEC_SUM=1 timeout 300 python tools/dev/decbench.py 4096 2>&1 | grep -v "amdgpu.ids\|^  segment"
done; } 2>&1 | tee $O/out.txt
