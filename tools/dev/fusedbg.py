import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from rplidar_ros2_driver_amd import Params, RplGpu, synth, abi
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
noise = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
n = 32000
batch = synth.make_batch(2031, B, n, noise_m=noise)
dev = torch.device("cuda:0")
d_nodes = torch.from_numpy(batch.view(np.uint8).reshape(B, n * 8)).to(dev)
d_len = torch.full((B,), n, dtype=torch.int32, device=dev)
cap = B * n
d_arena = torch.empty(cap, 4, dtype=torch.float32, device=dev)
d_cur = torch.zeros(1, dtype=torch.int64, device=dev)
items = (B + G - 1) // G
d_start = torch.zeros(items, dtype=torch.int64, device=dev)
d_np = torch.zeros(items, dtype=torch.int32, device=dev)
d_st = torch.zeros(items, dtype=torch.int32, device=dev)
d_dbg = torch.zeros(items, 16, dtype=torch.int64, device=dev)
rng = np.random.default_rng(2026)
motion = np.stack([[rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-0.3, 0.3), 0.1 / n] for _ in range(B)]).astype(np.float32)
ang = rng.uniform(-3, 3, B)
pose = np.stack([np.cos(ang), -np.sin(ang), rng.uniform(-2, 2, B), np.sin(ang), np.cos(ang), rng.uniform(-2, 2, B)], 1).astype(np.float32)
d_mo, d_po = torch.from_numpy(motion).to(dev), torch.from_numpy(pose).to(dev)
gpu = RplGpu(0, 32768, B)
lib = abi.load_library()
p = Params.defaults(clip_enable=1, range_min=0.15, range_max=40.0, voxel_enable=1, voxel_leaf=0.05)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); gpu.set_stream(st.cuda_stream)
def run():
    gpu.cloud_fused_voxel_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, G, p, d_mo.data_ptr(), d_po.data_ptr(),
                              d_arena.data_ptr(), cap, d_cur.data_ptr(), d_start.data_ptr(), d_np.data_ptr(), d_st.data_ptr())
for it in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st); run(); b.record(st); torch.cuda.synchronize()
    print("plain kernel ms", a.elapsed_time(b))
lib.rplgpu_debug_set_cycle_buffer(gpu._h, C.c_void_p(d_dbg.data_ptr()))
for it in range(2):
    d_dbg.zero_()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st); run(); b.record(st); torch.cuda.synchronize()
    print("dbg kernel ms", a.elapsed_time(b))
dbg = d_dbg.cpu().numpy()
nrec = dbg[:, 7] >> 40
dbg[:, 7] &= (1 << 40) - 1
print("records/item mean %.0f max %d" % (nrec.mean(), nrec.max()))
names = ["stream", "load+rowminmax", "select(store)", "rowscan", "scatter", "rank+permute", "heads+scan", "emit"]
for i, nm in enumerate(names):
    print("  %-16s mean %8.0f  p50 %8.0f  p99 %8.0f" % (nm, dbg[:, i].mean(), np.median(dbg[:, i]), np.percentile(dbg[:, i], 99)))
print("  total mean %.0f  cells/item %.0f" % (dbg[:, :8].sum(1).mean(), d_np.cpu().numpy().mean()))
