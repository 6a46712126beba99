#!/bin/bash
# tools/dev/rb.sh — developer aid: bench.py as the driver runs it + the forced single-rank N > 1 path
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out/rb; export RPL_SYNTH_CACHE=/tmp/rplc
( time timeout 900 python bench.py > gpurun_out/rb/bench.json 2> gpurun_out/rb/bench.err ) 2>&1 | grep real
tail -2 gpurun_out/rb/bench.err
RPL_BENCH_FORCE_DIST=1 timeout 600 python bench.py --cpu-seconds 0 --no-laserscan --no-variants --no-decode --no-single > gpurun_out/rb/dist.json 2> gpurun_out/rb/dist.err; tail -2 gpurun_out/rb/dist.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/rb/bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], "roofline", {k: d["roofline"][k] for k in ("frac", "frac_read", "frac_of_copy_rate", "kernel_ms_avg", "kernel_ms_min")})
print("variants", json.dumps(d.get("variants")))
print("c5", json.dumps(d.get("c5")))
print("single", json.dumps(d.get("single_scan_us")))
print("decode", json.dumps(d.get("decode")))
print("refgpu", json.dumps(d.get("reference_path_gpu")))
print("cpu", json.dumps(d.get("cpu_baseline"))[:900])
x = json.load(open("gpurun_out/rb/dist.json"))
print("dist", x["ms_per_step"], json.dumps(x["compute_only"]))
PY
