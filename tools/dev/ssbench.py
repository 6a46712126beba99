"""Developer aid: wall clock per call of the single-scan entry points (the bench's table alone)."""
import json, sys
sys.path.insert(0, ".")
import bench
from rplidar_ros2_driver_amd import RplGpu, Params
gpu = RplGpu(0, 32768, 16)
pv = Params.defaults(clip_enable=1, q_min=0, range_min=0.15, range_max=40.0, voxel_enable=1, voxel_leaf=0.05)
print(json.dumps(bench.single_scan_table(gpu, pv, 1234, 0.0), indent=1))
