"""Developer aid: the serialised-message assembly kernels on the C3 shape (4096 x 32 000)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from rplidar_ros2_driver_amd import Params, RplGpu, synth, abi
B, n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 32000
FID = "laser_frame"
dev = torch.device("cuda:0")
gpu = RplGpu(0, 32768, B)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); gpu.set_stream(st.cuda_stream)
batch = synth.make_batch(2026, 64, n)
d_nodes = torch.from_numpy(batch.view(np.uint8).reshape(64, n * 8)).to(dev).repeat(B // 64, 1).contiguous()
d_len = torch.full((B,), n, dtype=torch.int32, device=dev)
d_r = torch.empty(B, n, dtype=torch.float32, device=dev); d_i = torch.empty(B, n, dtype=torch.float32, device=dev)
d_cnt = torch.zeros(B, dtype=torch.int32, device=dev); d_st = torch.zeros(B, dtype=torch.int32, device=dev)
d_stamps = torch.zeros(B, 2, dtype=torch.int32, device=dev); d_dur = torch.full((B,), 0.1, dtype=torch.float64, device=dev)
d_ml = torch.zeros(B, dtype=torch.int32, device=dev)
def t(fn, reps=5):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); fn(); b.record(st); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts[1:])
p = Params.defaults(range_max=40.0)
ls = lambda: gpu.laserscan_batch_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, p, d_r.data_ptr(), d_i.data_ptr(), d_cnt.data_ptr())
t_ls = t(ls)
stride = abi.msg_laserscan_layout(len(FID), n).total_len
d_msgs = torch.empty(B, stride, dtype=torch.uint8, device=dev)
asm = lambda: gpu.laserscan_msgs_dev(d_r.data_ptr(), d_i.data_ptr(), n, d_cnt.data_ptr(), B, p, FID, d_stamps.data_ptr(), d_dur.data_ptr(), d_msgs.data_ptr(), stride, d_ml.data_ptr(), d_st.data_ptr())
t_asm = t(asm)
beams = int(d_cnt.sum().item()); nbytes = int(d_ml.to(torch.int64).sum().item())
print(f"laserscan A {t_ls:.3f} ms; LaserScan messages {t_asm:.3f} ms ({nbytes/1e9:.3f} GB out, {2*nbytes/t_asm/1e6:.0f} GB/s r+w, {beams/t_asm/1e6:.1f} Gbeams/s)")
del d_msgs, d_r, d_i
pv = Params.defaults(clip_enable=1, range_max=40.0, voxel_enable=1)
cap = B * 4096
d_arena = torch.empty(cap, 4, dtype=torch.float32, device=dev); d_cur = torch.zeros(1, dtype=torch.int64, device=dev)
d_start = torch.zeros(B, dtype=torch.int64, device=dev); d_np = torch.zeros(B, dtype=torch.int32, device=dev)
vx = lambda: gpu.cloud_arena_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, pv, d_arena.data_ptr(), cap, d_cur.data_ptr(), d_start.data_ptr(), d_np.data_ptr(), d_st.data_ptr())
t_vx = t(vx)
mx = int(d_np.max().item())
stride = (abi.msg_cloud_layout(len(FID), mx).total_len + 3) & ~3
d_msgs = torch.empty(B, stride, dtype=torch.uint8, device=dev)
asm = lambda: gpu.cloud_msgs_dev(d_arena.data_ptr(), 0, d_start.data_ptr(), d_np.data_ptr(), B, FID, d_stamps.data_ptr(), d_msgs.data_ptr(), stride, d_ml.data_ptr(), d_st.data_ptr())
t_asm = t(asm)
nbytes = int(d_ml.to(torch.int64).sum().item())
print(f"voxel arena {t_vx:.3f} ms; PointCloud2 messages {t_asm:.3f} ms ({nbytes/1e9:.3f} GB out, {2*nbytes/t_asm/1e6:.0f} GB/s r+w), status {int(d_st.max().item())}")
