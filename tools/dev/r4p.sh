#!/bin/bash
# round 4 final: rocprofv3 kernel trace + PMC passes of bench.py, the full default bench line, the long fuzz run
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/r4p; mkdir -p $O; export RPL_SYNTH_CACHE=/tmp/rplc
bash tools/prof.sh r04 > $O/prof.log 2>&1; tail -40 $R/gpurun_out/prof_r04/summary.txt
timeout 900 python bench.py > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$?"; tail -2 $O/bench_full.err
RPL_FUZZ_SEEDS=8000 timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 12 > $O/fuzz8000.log 2>&1; echo "fuzz rc=$?"; tail -3 $O/fuzz8000.log
lscpu | egrep 'Model name|^CPU\(s\)|Thread|Socket' > $O/host.txt; nproc >> $O/host.txt
