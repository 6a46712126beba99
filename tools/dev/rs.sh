#!/bin/bash
# developer aid: single-scan latency table under the three completion / copy modes
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
python - <<'PY'
import os, time, json, subprocess, sys
code = r'''
import sys, time, json
sys.path.insert(0, ".")
import numpy as np
from rplidar_ros2_driver_amd import Params, RplGpu, synth
gpu = RplGpu(0, 32768, 4)
pl = Params.defaults(range_max=40.0)
pv = Params.defaults(clip_enable=1, range_max=40.0, voxel_enable=1)
pin = gpu.host_alloc(1 << 20)
out = {}
for n in (360, 3200, 8192, 32000):
    one = synth.make_scan(1, n, n)
    row = {}
    for name, fn in (("laserscan", lambda: gpu.scan_to_laserscan(one, pl, 0.1)),
                     ("msg_pinned", lambda: gpu.scan_to_laserscan_msg(one, pl, 0.1, "laser_frame", 1, 2, out=pin)),
                     ("ascend", lambda: gpu.ascend(one.copy())),
                     ("voxel", lambda: gpu.scan_to_cloud(one, pv))):
        for _ in range(30): fn()
        t0 = time.perf_counter()
        for _ in range(300): fn()
        row[name] = round((time.perf_counter() - t0) / 300 * 1e6, 1)
    out[n] = row
print(json.dumps(out))
'''
for env in ({"RPLGPU_ZERO_COPY": "0"}, {"RPLGPU_ZERO_COPY": "1", "RPLGPU_SPIN_SYNC": "0"}, {"RPLGPU_ZERO_COPY": "1", "RPLGPU_SPIN_SYNC": "1"}):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True)
    print(env, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:])
PY
