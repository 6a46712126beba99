#!/bin/bash
# the chunked exchange pipeline of an N > 1 step (four chunks with peers) forced onto one rank: every --chunks value,
# both exchange modes; the gathered cloud must be the rank's own cloud whatever the chunking
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/${1:-exchange_chunks}; mkdir -p $O; export RPL_SYNTH_CACHE=/tmp/rplc
for x in allgather gather; do for c in 1 2 4 8; do
  RPL_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 20 --no-variants --no-decode --no-single --no-laserscan --cpu-seconds 0 --exchange $x --chunks $c > $O/bench_${x}_$c.json 2> $O/bench_${x}_$c.err; rc=$?
  python - <<P
import json
try:
    d=json.load(open("$O/bench_${x}_$c.json"))
    print("$x chunks=$c rc=$rc ms_per_step", d["ms_per_step"], "cells", d.get("cells_out_rank0"), "status", d.get("status_bits"), {k:d["compute_only"].get(k) for k in ("ms_per_step","exchange_only_ms","overlapped_ms","chunks","model_ms_per_step","gathered_bytes_per_rank")})
except Exception as e:
    print("$x chunks=$c rc=$rc FAILED", e); print(open("$O/bench_${x}_$c.err").read()[-1500:])
P
done; done
