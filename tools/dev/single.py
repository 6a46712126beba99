"""Developer aid: the single-scan host-buffer entry points in a loop (for rocprofv3 --kernel-trace)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from rplidar_ros2_driver_amd import Params, RplGpu, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32000
one = synth.make_scan(2026, 0, n)
gpu = RplGpu(0, 32768, 1)
p = Params.defaults(range_max=40.0)
pv = Params.defaults(clip_enable=1, range_max=40.0, voxel_enable=1)
pr = Params.defaults(clip_enable=1, range_max=40.0, voxel_enable=1, ror_enable=1)
work = one.copy()
def asc_reuse():
    work[:] = one
    gpu.ascend(work)
for name, fn in (("ascend(copy)", lambda: gpu.ascend(one.copy())), ("ascend(reused buffer)", asc_reuse), ("numpy copy alone", lambda: one.copy()), ("laserscan", lambda: gpu.scan_to_laserscan(one, p, 0.1)),
                 ("voxel", lambda: gpu.scan_to_cloud(one, pv)), ("ror+voxel", lambda: gpu.scan_to_cloud(one, pr))):
    for _ in range(10): fn()
    t0 = time.perf_counter()
    for _ in range(100): fn()
    print(f"{name}: {(time.perf_counter() - t0) / 100 * 1e6:.1f} us per call (n = {n})")
