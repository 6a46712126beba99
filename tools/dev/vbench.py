"""Developer aid: time rplgpu_cloud_arena_dev on a C3-shaped batch under the current environment
(RPLGPU_LIBRARY, VB_* ...).  One line per call:
  python tools/dev/vbench.py [B=4096] [reps=30] [noise_m=0] [kind=ring]"""
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from rplidar_ros2_driver_amd import Params, RplGpu, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
noise = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
kind = sys.argv[4] if len(sys.argv) > 4 else "ring"
n = 32000
dev = torch.device("cuda:0")
kw = {"kind": kind} if kind != "ring" else ({"noise_m": noise} if noise else {})
if os.environ.get("VB_R0MAX"):
    kw["r0_range"] = (1.0, float(os.environ["VB_R0MAX"]))
batch = synth.make_batch(2026, B, n, **kw)
d_nodes = torch.from_numpy(batch.view(np.uint8).reshape(B, n * 8)).to(dev)
d_len = torch.full((B,), n, dtype=torch.int32, device=dev)
cap = B * (8192 if kind == "ring" and not noise else n)
d_arena = torch.empty(cap, 4, dtype=torch.float32, device=dev)
d_cur = torch.zeros(1, dtype=torch.int64, device=dev)
d_start = torch.zeros(B, dtype=torch.int64, device=dev)
d_np = torch.zeros(B, dtype=torch.int32, device=dev)
d_st = torch.zeros(B, dtype=torch.int32, device=dev)
p = Params.defaults(clip_enable=1, q_min=0, range_min=0.15, range_max=40.0, voxel_enable=1, voxel_leaf=0.05)
gpu = RplGpu(device=0, max_samples_per_scan=32768, max_batch=B)
if os.environ.get("VB_AGG"):  # 0 auto, 1 plain, 2 two-class (rplgpu_set_voxel_aggregation)
    gpu.set_voxel_aggregation(int(os.environ["VB_AGG"]))
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
gpu.set_stream(stream.cuda_stream)


def step():
    gpu.cloud_arena_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, p, d_arena.data_ptr(), cap,
                        d_cur.data_ptr(), d_start.data_ptr(), d_np.data_ptr(), d_st.data_ptr())


for _ in range(5):
    step()
torch.cuda.synchronize()
best = 1e9
tot = 0.0
for _ in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(reps):
        step()
    b.record(stream)
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    best = min(best, ms)
    tot += ms
cells = int(d_cur.item())
algo = 8 * B * n + 16 * cells
print(f"lib={Path(os.environ.get('RPLGPU_LIBRARY', 'default')).name} B={B} noise={noise} kind={kind} "
      f"ms_best={best:.4f} ms_avg={tot / 3:.4f} cells={cells} status={int(d_st.max().item())} "
      f"frac={algo / (best * 1e-3) / 8e12:.3f}")
