"""Developer aid: rplgpu_ascend_batch_dev on uniform angle words over batch shapes (per-scan cost vs streaming rate).
  python tools/dev/ascscale.py"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from rplidar_ros2_driver_amd import RplGpu, synth  # noqa: E402

dev = torch.device("cuda:0")
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
for B, n in ((2048, 32000), (4096, 32000), (8192, 32000), (8192, 16000), (16384, 8000), (32768, 4000)):
    gpu = RplGpu(device=0, max_samples_per_scan=32768, max_batch=B)
    gpu.set_stream(stream.cuda_stream)
    d_len = torch.full((B,), n, dtype=torch.int32, device=dev)
    d_st = torch.zeros(B, dtype=torch.int32, device=dev)
    base = synth.make_batch(2037, min(B, 1024), n)
    vb = np.concatenate([base] * (B // len(base)))
    d_w = torch.from_numpy(vb.view(np.uint8).reshape(B, n * 8)).to(dev)
    ts = []
    for it in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        gpu.ascend_batch_dev(d_w.data_ptr(), n, d_len.data_ptr(), B, d_st.data_ptr())
        b.record(stream)
        torch.cuda.synchronize(dev)
        ts.append(a.elapsed_time(b))
    ms = min(ts[1:])
    print(f"B={B:6d} n={n:6d} ms={ms:.4f} GB/s={8 * B * n / ms / 1e6:.0f} frac={8 * B * n / (ms * 1e-3) / 8e12:.3f} status={int(d_st.max())}")
    gpu.close()
    del d_w
