"""Developer aid: k_laserscan_a on the bench batch (4096 x 32768), time per launch and — with a
library built with -DRPL_LS_DBG — the phase clocks of workgroup thread 0 (100 MHz ticks)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from rplidar_ros2_driver_amd import RplGpu, Params, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
pad = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # extra samples between scans (stride = n + pad)
n = 32768
dev = torch.device("cuda:0")
batch = synth.make_batch(1234, min(B, 256), n)
d_nodes = torch.from_numpy(batch.view(np.uint8).reshape(-1, n * 8)).to(dev).repeat((B + 255) // 256, 1)[:B].contiguous()
if pad:
    d_pad = torch.zeros(B, (n + pad) * 8, dtype=torch.uint8, device=dev)
    d_pad[:, :n * 8] = d_nodes
    d_nodes = d_pad
ns = n + pad
d_len = torch.full((B,), n, dtype=torch.int32, device=dev)
gpu = RplGpu(0, 32768, B)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); gpu.set_stream(st.cuda_stream)
d_st = torch.zeros(B, dtype=torch.int32, device=dev)
gpu.ascend_batch_dev(d_nodes.data_ptr(), ns, d_len.data_ptr(), B, d_st.data_ptr())
d_r = torch.zeros(B, ns, dtype=torch.float32, device=dev)
d_i = torch.zeros(B, ns, dtype=torch.float32, device=dev)
d_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
pl = Params.defaults(range_max=40.0)
run = lambda: gpu.laserscan_batch_dev(d_nodes.data_ptr(), ns, d_len.data_ptr(), B, pl, d_r.data_ptr(), d_i.data_ptr(), d_cnt.data_ptr())
run(); torch.cuda.synchronize()
ts = []
for _ in range(8):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st); run(); b.record(st); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
cnt = d_cnt.cpu().numpy()
print(f"laserscan_a B={B} pad={pad}: min {min(ts):.4f} ms  median {sorted(ts)[len(ts)//2]:.4f} ms   beams mean {cnt.mean():.0f}")
ph = d_r[:, ns - 8:ns - 2].cpu().numpy().astype(np.float64)
if np.isfinite(ph).all() and ph.max() > 0 and ph.max() < 1e7:
    print("phase ticks (100 MHz) load+count, convert, win0 atomics, win0 flush, win1 atomics, win1 flush:")
    print("  mean", ph.mean(0).round(0), " sum", ph.sum(1).mean().round(0))
    print("  p10 ", np.percentile(ph, 10, axis=0).round(0))
    print("  p90 ", np.percentile(ph, 90, axis=0).round(0))
