"""Developer aid: phase cycle breakdown of k_cloud_voxel on the bench workload."""
import ctypes as C
import os
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from rplidar_ros2_driver_amd import Params, RplGpu, synth, abi

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(os.environ.get('RPL_VOXDBG_N', '32000')) if True else 32000
NOISE = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
import os
R0MAX = float(os.environ.get('RPL_VOXDBG_R0MAX', '30'))
KIND = os.environ.get('RPL_VOXDBG_KIND', 'ring')
batch = synth.make_batch(2026, B, n, kind='uniform') if KIND == 'uniform' else \
    synth.make_batch(2026, B, n, noise_m=NOISE, r0_range=(1.0, R0MAX))
STRIDE = n if KIND == 'uniform' or NOISE >= 0.02 else 8192
dev = torch.device("cuda:0")
d_nodes = torch.from_numpy(batch.view(np.uint8).reshape(B, n * 8)).to(dev)
d_len = torch.full((B,), n, dtype=torch.int32, device=dev)
d_xyzi = torch.empty(B, STRIDE, 4, dtype=torch.float32, device=dev)
d_np = torch.zeros(B, dtype=torch.int32, device=dev)
d_st = torch.zeros(B, dtype=torch.int32, device=dev)
d_dbg = torch.zeros(B, 16, dtype=torch.int64, device=dev)
gpu = RplGpu(0, 32768, B)
lib = abi.load_library()
lib.rplgpu_debug_set_cycle_buffer(gpu._h, C.c_void_p(d_dbg.data_ptr()))
p = Params.defaults(clip_enable=1, range_max=40.0, voxel_enable=1)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); gpu.set_stream(st.cuda_stream)
for it in range(3):
    d_dbg.zero_()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    gpu.cloud_batch_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, p, d_xyzi.data_ptr(), STRIDE,
                        d_np.data_ptr(), d_st.data_ptr())
    b.record(st); torch.cuda.synchronize()
    print("kernel ms", a.elapsed_time(b))
print("fast_div flags", lib.rplgpu_debug_fast_div(gpu._h))
dbg = d_dbg.cpu().numpy()
nrec = dbg[:, 7] >> 40
dbg[:, 7] &= (1 << 40) - 1
print("records/scan mean %.0f p50 %.0f p90 %.0f p99 %.0f max %d" % (nrec.mean(), np.median(nrec), np.percentile(nrec, 90), np.percentile(nrec, 99), nrec.max()))
names = ["stream", "load+rowminmax", "select(store)", "rowscan", "scatter", "rank+permute", "heads+scan", "emit"]
for i, nm in enumerate(names):
    print("  %-16s mean %8.0f  p50 %8.0f  p99 %8.0f" % (nm, dbg[:, i].mean(), np.median(dbg[:, i]), np.percentile(dbg[:, i], 99)))
print("  total mean %.0f" % dbg[:, :8].sum(1).mean())
nb = np.maximum(dbg[:, 12], 1)
print("  wave 0, cycles per block: issue loads %.0f  wait entries + arithmetic %.0f  block pass %.0f  loop overhead %.0f  (blocks/scan %.1f)" % (
    (dbg[:, 8] / nb).mean(), (dbg[:, 9] / nb).mean(), (dbg[:, 10] / nb).mean(), (dbg[:, 11] / nb).mean(), dbg[:, 12].mean()))
npts = d_np.cpu().numpy()
print("cells mean", npts.mean(), "status", int(d_st.max()))
