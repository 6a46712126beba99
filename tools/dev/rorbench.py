"""Developer aid: k_ror_mask on the bench's config-5 batch (1 cm range noise): time of the
E5 + plain-cloud call and — with a library built with -DRPL_ROR_DBG — the phase clocks of thread 0."""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from rplidar_ros2_driver_amd import Params, RplGpu, synth, abi

B, n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 32000
NOISE = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
batch = synth.make_batch(2031, B, n, noise_m=NOISE)
dev = torch.device("cuda:0")
d_nodes = torch.from_numpy(batch.view(np.uint8).reshape(B, n * 8)).to(dev)
d_len = torch.full((B,), n, dtype=torch.int32, device=dev)
d_xyzi = torch.empty(B, n, 4, dtype=torch.float32, device=dev)
d_np = torch.zeros(B, dtype=torch.int32, device=dev)
d_st = torch.zeros(B, dtype=torch.int32, device=dev)
d_dbg = torch.zeros(B, 16, dtype=torch.int64, device=dev)
gpu = RplGpu(0, 32768, B)
lib = abi.load_library()
lib.rplgpu_debug_set_cycle_buffer(gpu._h, C.c_void_p(d_dbg.data_ptr()))
st = torch.cuda.Stream(); torch.cuda.set_stream(st); gpu.set_stream(st.cuda_stream)
res = {}
for ror in (0, 1):
    p = Params.defaults(clip_enable=1, range_max=40.0, ror_enable=ror, ror_radius=0.10, ror_min_neighbors=2)
    ts = []
    for it in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        gpu.cloud_batch_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, p, d_xyzi.data_ptr(), n,
                            d_np.data_ptr(), d_st.data_ptr())
        b.record(st); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    res[ror] = min(ts)
    print(f"ror={ror}: plain cloud call {min(ts):.3f} ms, points kept {int(d_np.sum().item())}")
print(f"k_ror_mask ~ {res[1] - res[0]:.3f} ms for {B} scans")
dbg = d_dbg.cpu().numpy()
if dbg[:, :4].max() > 0:
    for i, nm in enumerate(["stage 1", "stage 1b", "rows", "stage 2"]):
        print("  %-9s mean %8.0f  p50 %8.0f  p99 %8.0f" % (nm, dbg[:, i].mean(), np.median(dbg[:, i]), np.percentile(dbg[:, i], 99)))
    print("  prologue mean %.0f, mask write-out mean %.0f, whole workgroup mean %.0f" % (
        dbg[:, 6].mean(), dbg[:, 7].mean(), (dbg[:, 9] - dbg[:, 8]).mean()))
    hw = dbg[:, 10]
    cu = ((hw >> 8) & 0xF) | (((hw >> 13) & 0x7) << 4) | (((hw >> 16) & 0xF) << 7)  # cu, sh/se bits (rough id)
    gaps = []
    for c in np.unique(cu):
        sel = np.nonzero(cu == c)[0]
        o = sel[np.argsort(dbg[sel, 8])]
        gaps += list(dbg[o[1:], 8] - dbg[o[:-1], 9])
    if gaps:
        g = np.array(gaps, dtype=np.float64)
        print("  gap between a workgroup's end and the next one's entry on the same CU id: median %.0f mean %.0f (n=%d, %d ids)" % (
            np.median(g), g.mean(), len(g), len(np.unique(cu))))
    print("  unsettled after stage 1: mean %.1f p99 %.0f; after 1b: mean %.1f p99 %.0f" % (
        dbg[:, 4].mean(), np.percentile(dbg[:, 4], 99), dbg[:, 5].mean(), np.percentile(dbg[:, 5], 99)))
