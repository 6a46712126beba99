"""Developer aid: rplgpu_cloud_batch_dev WITHOUT the voxel grid (E1 + E2 -> 16-byte points) on the C3 batch.
  python tools/dev/cldbench.py [B=4096]"""
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from rplidar_ros2_driver_amd import Params, RplGpu, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = 32000
dev = torch.device("cuda:0")
batch = synth.make_batch(2026, B, n)
d_nodes = torch.from_numpy(batch.view(np.uint8).reshape(B, n * 8)).to(dev)
d_len = torch.full((B,), n, dtype=torch.int32, device=dev)
d_xyzi = torch.empty(B, n, 4, dtype=torch.float32, device=dev)
d_np = torch.zeros(B, dtype=torch.int32, device=dev)
d_st = torch.zeros(B, dtype=torch.int32, device=dev)
gpu = RplGpu(device=0, max_samples_per_scan=32768, max_batch=B)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
gpu.set_stream(stream.cuda_stream)
for name, p in (("plain", Params.defaults(clip_enable=1, range_max=40.0)),
                ("q48", Params.defaults(clip_enable=1, q_min=48, range_max=40.0))):
    def step():
        gpu.cloud_batch_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, p, d_xyzi.data_ptr(), n,
                            d_np.data_ptr(), d_st.data_ptr())
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(10):
            step()
        b.record(stream)
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 10)
    pts = int(d_np.to(torch.int64).sum().item())
    algo = 8 * B * n + 16 * pts
    print(f"lib={Path(os.environ.get('RPLGPU_LIBRARY', 'default')).name} {name}: ms={best:.4f} points={pts} "
          f"frac={algo / (best * 1e-3) / 8e12:.3f} GB/s={algo / best / 1e6:.0f}")
