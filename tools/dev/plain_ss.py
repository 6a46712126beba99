import sys, time
import numpy as np
sys.path.insert(0, ".")
from rplidar_ros2_driver_amd import RplGpu, Params, synth
gpu = RplGpu(0, 32768, 16)
pc = Params.defaults(clip_enable=1, q_min=0, range_min=0.15, range_max=40.0)
pb = Params.defaults(range_max=40.0, scan_processing=0)
for n in (360, 3200, 8192, 32000):
    one = synth.make_scan(1234, 9000 + n, n)
    row = {}
    for name, fn in (("plain_cloud", lambda: gpu.scan_to_cloud(one, pc)), ("laserscan_mode_b", lambda: gpu.scan_to_laserscan(one, pb, 0.1))):
        for _ in range(20): fn()
        t0 = time.perf_counter()
        for _ in range(200): fn()
        row[name] = round((time.perf_counter() - t0) / 200 * 1e6, 1)
    print(n, row)
