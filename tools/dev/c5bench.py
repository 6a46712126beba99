"""Developer aid: config 5 (E5 + E4) on the bench's C5 batch, E5 inside the voxel kernel against the two
kernels (rplgpu_set_ror_mode), per-scan arena and one grid per 8 sensors (E8).  Same box, alternating:
  python tools/dev/c5bench.py [B=4096] [reps=10]"""
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from rplidar_ros2_driver_amd import Params, RplGpu, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n = 32000
dev = torch.device("cuda:0")
batch = synth.make_batch(2026 + 5, B, n, noise_m=0.01, invalid_p=float(os.environ.get("C5_INVALID", "0.10")))
d_nodes = torch.from_numpy(batch.view(np.uint8).reshape(B, n * 8)).to(dev)
d_len = torch.full((B,), n, dtype=torch.int32, device=dev)
cap = B * n
d_arena = torch.empty(cap, 4, dtype=torch.float32, device=dev)
d_cur = torch.zeros(1, dtype=torch.int64, device=dev)
d_start = torch.zeros(B, dtype=torch.int64, device=dev)
d_np = torch.zeros(B, dtype=torch.int32, device=dev)
d_st = torch.zeros(B, dtype=torch.int32, device=dev)
p = Params.defaults(clip_enable=1, q_min=0, range_min=0.15, range_max=40.0, voxel_enable=1, voxel_leaf=0.05,
                    ror_enable=1, ror_radius=0.10, ror_min_neighbors=2)
rng = np.random.default_rng(2026)
motion = np.stack([[rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-0.3, 0.3), 0.1 / n]
                   for _ in range(B)]).astype(np.float32)
ang = rng.uniform(-3, 3, B)
pose2d = np.stack([np.cos(ang), -np.sin(ang), rng.uniform(-2, 2, B), np.sin(ang), np.cos(ang),
                   rng.uniform(-2, 2, B)], 1).astype(np.float32)
d_mo, d_po = torch.from_numpy(motion).to(dev), torch.from_numpy(pose2d).to(dev)
gpu = RplGpu(device=0, max_samples_per_scan=32768, max_batch=B)
if os.environ.get("VB_AGG"):
    gpu.set_voxel_aggregation(int(os.environ["VB_AGG"]))
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
gpu.set_stream(stream.cuda_stream)


def arena():
    gpu.cloud_arena_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, p, d_arena.data_ptr(), cap,
                        d_cur.data_ptr(), d_start.data_ptr(), d_np.data_ptr(), d_st.data_ptr())


def fused():
    gpu.cloud_fused_voxel_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, 8, p, d_mo.data_ptr(),
                              d_po.data_ptr(), d_arena.data_ptr(), cap, d_cur.data_ptr(), d_start.data_ptr(),
                              d_np.data_ptr(), d_st.data_ptr())


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(reps):
            fn()
        b.record(stream)
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / reps)
    return best


p_ror = p
p = Params.defaults(clip_enable=1, q_min=0, range_min=0.15, range_max=40.0, voxel_enable=1, voxel_leaf=0.05)
if not os.environ.get("C5_ONLY"):
    print(f"lib={Path(os.environ.get('RPLGPU_LIBRARY', 'default')).name} without E5: arena {timed(arena):.4f} ms, "
          f"fused grid {timed(fused):.4f} ms", flush=True)
    p = Params.defaults(clip_enable=1, q_min=1, range_min=0.15, range_max=40.0, voxel_enable=1, voxel_leaf=0.05)
    print(f"  without E5, q_min 1 (quality test + FILL pass): arena {timed(arena):.4f} ms", flush=True)
p = p_ror
for rnd in range(int(os.environ.get("C5_ROUNDS", "2"))):
    for name, fn in (("c5 arena", arena), ("fused grid x8", fused)):
        if os.environ.get("C5_ONLY") and os.environ["C5_ONLY"] not in name:
            continue
        row = []
        for mode in (0, 1):
            gpu.set_ror_mode(mode)
            ms = timed(fn)
            cells = int(d_cur.item())
            listed = gpu.debug_ror_listed() if mode == 0 else -1
            row.append(f"{'inside' if mode == 0 else 'two kernels'} {ms:.4f} ms cells {cells} "
                       f"frac {(8 * B * n + 16 * cells) / (ms * 1e-3) / 8e12:.3f}"
                       + (f" listed {listed}" if mode == 0 else ""))
        print(f"{name}: " + " | ".join(row), flush=True)
