"""Developer aid: timeline of the pipelined two-kernel voxel path (RPLGPU_VOXEL_PATH=two): when was an item's
last region written (k_voxel_runs), when did a k_voxel_cells workgroup begin to wait for it / see it ready.
100 MHz wall clock; prints, per item quantile, the times relative to the first stamp."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from rplidar_ros2_driver_amd import Params, RplGpu, synth, abi  # noqa: E402

B, n = 4096, 32000
dev = torch.device("cuda:0")
batch = synth.make_batch(2026, B, n)
d_nodes = torch.from_numpy(batch.view(np.uint8).reshape(B, n * 8)).to(dev)
d_len = torch.full((B,), n, dtype=torch.int32, device=dev)
cap = B * 8192
d_arena = torch.empty(cap, 4, dtype=torch.float32, device=dev)
d_cur = torch.zeros(1, dtype=torch.int64, device=dev)
d_start = torch.zeros(B, dtype=torch.int64, device=dev)
d_np = torch.zeros(B, dtype=torch.int32, device=dev)
d_st = torch.zeros(B, dtype=torch.int32, device=dev)
p = Params.defaults(clip_enable=1, q_min=0, range_min=0.15, range_max=40.0, voxel_enable=1, voxel_leaf=0.05)
gpu = RplGpu(device=0, max_samples_per_scan=32768, max_batch=B)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
gpu.set_stream(stream.cuda_stream)
lib = abi.load_library()
d_dbg = torch.zeros(B, 16, dtype=torch.int64, device=dev)


def step():
    gpu.cloud_arena_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, p, d_arena.data_ptr(), cap,
                        d_cur.data_ptr(), d_start.data_ptr(), d_np.data_ptr(), d_st.data_ptr())


for _ in range(3):
    step()
torch.cuda.synchronize()
lib.rplgpu_debug_set_cycle_buffer(gpu._h, C.c_void_p(d_dbg.data_ptr()))
d_dbg.zero_()
step()
torch.cuda.synchronize()
lib.rplgpu_debug_set_cycle_buffer(gpu._h, C.c_void_p(0))
d = d_dbg.cpu().numpy().astype(np.float64)
t0 = d[:, [0, 1, 3]][d[:, [0, 1, 3]] > 0].min()
us = lambda x: (x - t0) / 100.0
print("item      produced(us)  wait_begin(us)  seen_ready(us)")
for i in [0, 1, 255, 256, 512, 1024, 2048, 3072, 4095]:
    print("%5d  %12.1f  %14.1f  %14.1f" % (i, us(d[i, 3]), us(d[i, 0]), us(d[i, 1])))
print("last produced %.1f us; last seen ready %.1f us; consumer idle before its first item ready: %.1f us" %
      (us(d[:, 3].max()), us(d[:, 1].max()), us(d[:256, 1].min())))
lag = (d[:, 1] - d[:, 3]) / 100.0
print("ready -> picked up lag: mean %.1f  p50 %.1f  p90 %.1f  max %.1f us" % (lag.mean(), np.median(lag), np.percentile(lag, 90), lag.max()))
wait = (d[:, 1] - d[:, 0]) / 100.0
print("consumer waited for the producer: mean %.1f p50 %.1f us; items it waited > 1 us for: %d" % (wait.mean(), np.median(wait), int((wait > 1).sum())))
