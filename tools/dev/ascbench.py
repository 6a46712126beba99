"""Developer aid: rplgpu_ascend_batch_dev over angle-jitter regimes (1024 x 32 000 samples).
  python tools/dev/ascbench.py [B=1024]"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from rplidar_ros2_driver_amd import RplGpu, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = 32000
dev = torch.device("cuda:0")
gpu = RplGpu(device=0, max_samples_per_scan=32768, max_batch=B)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
gpu.set_stream(stream.cuda_stream)
d_len = torch.full((B,), n, dtype=torch.int32, device=dev)
d_st = torch.zeros(B, dtype=torch.int32, device=dev)
import ctypes as C
from rplidar_ros2_driver_amd import abi  # noqa: E402
_lib = abi.load_library()
JITS = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else (0, 1, 2, 3, 6, 10, 20, 40, 64, 128, 300)
for jit in JITS:
    vb = synth.make_batch(2037, B, n, jitter=jit)
    d_v = torch.from_numpy(vb.view(np.uint8).reshape(B, n * 8)).to(dev)
    d_w = d_v.clone()
    ts = []
    for it in range(6):
        d_w.copy_(d_v)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        gpu.ascend_batch_dev(d_w.data_ptr(), n, d_len.data_ptr(), B, d_st.data_ptr())
        b.record(stream)
        torch.cuda.synchronize(dev)
        ts.append(a.elapsed_time(b))
    ms = min(ts[1:])
    x, y = d_v.view(B, n, 8), d_w.view(B, n, 8)
    changed = int((x != y).any(dim=2).sum().item())
    reord = int((x[:, :, 2:6] != y[:, :, 2:6]).any(dim=2).any(dim=1).sum().item())
    frac = (8 * B * n + 8 * changed) / (ms * 1e-3) / 8e12
    nsort = C.c_uint32(0)
    try:
        _lib.rplgpu_debug_ascend_sorted.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        _lib.rplgpu_debug_ascend_sorted(gpu._h, C.byref(nsort))
        ns = nsort.value
    except AttributeError:
        ns = -1
    print(f"jitter={jit:3d} ms={ms:.4f} rewritten={changed / (B * n):.4f} scans_reordered={reord / B:.3f} "
          f"to_sort_kernel={ns} frac={frac:.3f}")
