#!/bin/bash
# tools/dev/hqprof.sh [lib suffixes...] — the HQ decoder (k_decode<0x83>): tools/dev/decbench.py time, then HBM traffic
# (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, counter-only passes; FETCH_SIZE x 2 on gfx950), per library build
R=${GRAFT_REPO_ROOT:-$PWD}; L=$R/rplidar_ros2_driver_amd/lib; cd $R; export TMPDIR=/tmp
for v in "" "$@"; do
  echo "== lib$v"
  DEC_ONLY=0x83 RPLGPU_LIBRARY=$L/librplgpu$v.so python $R/tools/dev/decbench.py 2>&1 | grep "ans 0x83"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/hqp; DEC_ONLY=0x83 RPLGPU_LIBRARY=$L/librplgpu$v.so timeout 200 rocprofv3 --pmc $c --output-format csv -d /tmp/hqp -o p -- python $R/tools/dev/decbench.py > /dev/null 2>&1
    python - <<PY
import csv,glob,collections
f=glob.glob("/tmp/hqp/**/*counter_collection.csv",recursive=True)
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if r["Counter_Name"]=="$c": agg[r["Kernel_Name"][:50]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    if "k_decode" in k: print("  $c",k,len(v),"mean KiB %.0f"%(sum(v)/len(v)))
PY
  done
done
