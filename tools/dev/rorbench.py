"""Developer aid: E5 radius-outlier removal + voxel on a batch (BASELINE config 5 shape)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from rplidar_ros2_driver_amd import Params, RplGpu, synth
B, n = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 32000
batch = synth.make_batch(2026, B, n, noise_m=0.01)
dev = torch.device("cuda:0")
gpu = RplGpu(0, 32768, B)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); gpu.set_stream(st.cuda_stream)
d_nodes = torch.from_numpy(batch.view(np.uint8).reshape(B, n * 8)).to(dev)
d_len = torch.full((B,), n, dtype=torch.int32, device=dev)
d_xyzi = torch.empty(B, 8192, 4, dtype=torch.float32, device=dev)
d_np = torch.zeros(B, dtype=torch.int32, device=dev); d_st = torch.zeros(B, dtype=torch.int32, device=dev)
for name, p in (("voxel", Params.defaults(clip_enable=1, range_max=40.0, voxel_enable=1)),
                ("ror+voxel", Params.defaults(clip_enable=1, range_max=40.0, voxel_enable=1, ror_enable=1)),
                ("ror+cloud", Params.defaults(clip_enable=1, range_max=40.0, ror_enable=1))):
    if name == "ror+cloud":
        d_xyzi = torch.empty(B, n, 4, dtype=torch.float32, device=dev)
    stride = d_xyzi.shape[1]
    ts = []
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        gpu.cloud_batch_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, p, d_xyzi.data_ptr(), stride, d_np.data_ptr(), d_st.data_ptr())
        b.record(st); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print(f"{name}: {min(ts[1:]):.3f} ms for {B} scans -> {B*n/min(ts[1:])/1e6:.1f} Gpts/s, points {int(d_np.sum())}, status {int(d_st.max())}")
