#!/bin/bash
# tools/gpu_round.sh <tag> — one gpurun call: GPU parity tests, bench line, rocprofv3 profile
# (kernel trace + PMC passes) and the voxel phase-cycle breakdown.  Everything lands in
# gpurun_out/<tag>/.
set -u
TAG=${1:-r}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export RPL_SYNTH_CACHE=/tmp/rplc
cd $R
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
  echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
  tail -3 $OUT/pytest_gpu.log
fi
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; cat $OUT/bench.json; tail -3 $OUT/bench.err
if [ "${SKIP_PROF:-0}" != "1" ]; then
  bash tools/prof.sh $TAG > /dev/null 2>&1
  cat $R/gpurun_out/prof_$TAG/summary.txt
fi
if [ "${SKIP_DBG:-0}" != "1" ]; then
  timeout 300 python tools/voxdbg.py 1024 > $OUT/voxdbg.txt 2>&1; cat $OUT/voxdbg.txt
fi
lscpu | egrep 'Model name|^CPU\(s\)|Thread|Socket' > $OUT/host.txt; nproc >> $OUT/host.txt; cat $OUT/host.txt
