#!/bin/bash
# round-2 GPU call 1: full GPU suite with the new (ring) build, A/B of the streaming loop,
# regime baselines, bench line
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export RPL_SYNTH_CACHE=/tmp/rplc
OUT=$R/gpurun_out/r2_1; mkdir -p $OUT
L=$R/rplidar_ros2_driver_amd/lib
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
for i in 1 2; do for v in A B; do echo -n "$v: "; RPLGPU_LIBRARY=$L/librplgpu_$v.so timeout 120 python tools/voxdbg.py 2048 2>&1 | egrep "kernel ms|stream|total mean" | tail -3 | tr '\n' ' '; echo; done; done | tee $OUT/ab.txt
for v in A B; do echo "== $v noisy"; RPLGPU_LIBRARY=$L/librplgpu_$v.so timeout 120 python tools/voxdbg.py 256 0.01 2>&1 | tail -12; done > $OUT/noisy.txt 2>&1
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["kernel_ms_min"], d.get("c5_ror_voxel_ms"), d.get("single_scan_us"))
PY
tail -3 $OUT/bench.err
