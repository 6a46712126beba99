"""Developer aid: the secondary paths on the C3 shape (Mode B, unsorted input, plain cloud)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from rplidar_ros2_driver_amd import Params, RplGpu, synth
B, n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 32000
dev = torch.device("cuda:0")
gpu = RplGpu(0, 32768, B)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); gpu.set_stream(st.cuda_stream)
d_len = torch.full((B,), n, dtype=torch.int32, device=dev)
d_r = torch.empty(B, n, dtype=torch.float32, device=dev); d_i = torch.empty(B, n, dtype=torch.float32, device=dev)
d_cnt = torch.zeros(B, dtype=torch.int32, device=dev); d_st = torch.zeros(B, dtype=torch.int32, device=dev)
d_xyzi = torch.empty(B, n, 4, dtype=torch.float32, device=dev); d_np = torch.zeros(B, dtype=torch.int32, device=dev)
def t(fn, reps=4):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); fn(); b.record(st); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts[1:])
for tag, kw in (("sorted", {}), ("rotated+jitter", dict(rotate=True, jitter=3))):
    batch = synth.make_batch(2026, 64, n, **kw)
    d_nodes = torch.from_numpy(batch.view(np.uint8).reshape(64, n * 8)).to(dev).repeat(B // 64, 1).contiguous()
    work = d_nodes.clone()
    def asc():
        work.copy_(d_nodes)
        gpu.ascend_batch_dev(work.data_ptr(), n, d_len.data_ptr(), B, d_st.data_ptr())
    tc = t(lambda: work.copy_(d_nodes))
    print(f"[{tag}] ascend {t(asc) - tc:.3f} ms (copy {tc:.3f})", end="; ")
    for mode in (1, 0):
        p = Params.defaults(range_max=40.0, scan_processing=mode)
        ms = t(lambda: gpu.laserscan_batch_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, p, d_r.data_ptr(), d_i.data_ptr(), d_cnt.data_ptr()))
        print(f"laserscan mode {'A' if mode else 'B'} {ms:.3f} ms", end="; ")
    p = Params.defaults(clip_enable=1, range_max=40.0)
    ms = t(lambda: gpu.cloud_batch_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, p, d_xyzi.data_ptr(), n, d_np.data_ptr(), d_st.data_ptr()))
    print(f"plain cloud {ms:.3f} ms  ({B} scans)")
