#!/bin/bash
# end of round 4: the whole GPU suite (default, then with the two-kernel path forced and pipelined), smoke, profile, bench
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/r4s; mkdir -p $O; export RPL_SYNTH_CACHE=/tmp/rplc
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
RPLGPU_VOXEL_PATH=two RPLGPU_VOXEL_PIPE=3 timeout 1200 python -m pytest tests -m gpu -x -q -k "not node_patch and not gpu2_rccl" > $O/pytest_two_pipe.log 2>&1; echo "two+pipe rc=$?"; tail -3 $O/pytest_two_pipe.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/prof.sh r04b > $O/prof.log 2>&1; grep -A3 "k_cloud_voxel" $R/gpurun_out/prof_r04b/summary.txt | head -8; tail -1 $R/gpurun_out/prof_r04b/summary.txt
cp profiles/traffic.json $O/traffic_before.json; cp $R/gpurun_out/prof_r04b/traffic.json profiles/traffic.json
timeout 900 python bench.py > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$?"
python - <<PY
import json
l=json.loads(open("$O/bench_full.json").read().strip().splitlines()[-1])
print({k:l[k] for k in ("value","ms_per_step")}, {k:l["roofline"][k] for k in ("frac","kernel_ms_avg","kernel_ms_min","traffic","traffic_source")})
PY
