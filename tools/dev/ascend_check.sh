#!/bin/bash
# the ascend GPU tests + tools/dev/ascbench.py (jitter regimes at 1024 and 4096 scans) in one call
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out/${1:-ascend_check}; mkdir -p $O; export RPL_SYNTH_CACHE=/tmp/rplc
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_node_patch.py -m gpu -x -q -k "ascend or fuzz or node" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
{ timeout 300 python tools/dev/ascbench.py 1024; timeout 300 python tools/dev/ascbench.py 4096; } 2>&1 | grep jitter= | tee $O/ascbench.txt
