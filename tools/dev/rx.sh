#!/bin/bash
# developer aid: the forced single-rank N > 1 path, native exchange and torch.distributed fallback
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out/rx; export RPL_SYNTH_CACHE=/tmp/rplc
for m in native torch; do
  [ $m = torch ] && export RPL_EXCHANGE=torch
  RPL_BENCH_FORCE_DIST=1 timeout 600 python bench.py --cpu-seconds 0 --no-laserscan --no-variants --no-decode --no-single > gpurun_out/rx/$m.json 2> gpurun_out/rx/$m.err; tail -2 gpurun_out/rx/$m.err
  python -c "
import json; x = json.load(open('gpurun_out/rx/$m.json')); print('$m', x['ms_per_step'], json.dumps(x['compute_only']))"
done
