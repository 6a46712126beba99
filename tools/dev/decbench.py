"""Developer aid: throughput of the decode stage (k_decode / k_segment / k_scans_to_batch)."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from rplidar_ros2_driver_amd import RplGpu, capsules as cp

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
gpu = RplGpu(0, 32768, 16)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); gpu.set_stream(st.cuda_stream)
import os
ONLY = int(os.environ.get('DEC_ONLY', '0'), 0)
for ans, nf in ((0x85, 801), (0x82, 1001), (0x84, 334), (0x86, 501), (0x83, 334), (0x81, 4000)):
    if ONLY and ans != ONLY:
        continue
    S, npf = cp.FRAME_SIZE[ans], cp.NODES_PER_FRAME[ans]
    uniq = 16
    base = np.stack([cp.make_stream(ans, nf, 10 + s, payload=os.environ.get('DEC_PAYLOAD', 'ring'), frames_per_rev=nf / 2.0 + 0.3) for s in range(uniq)])
    buf = torch.from_numpy(base).to(dev).repeat((B + uniq - 1) // uniq, 1)[:B].contiguous()
    d_nf = torch.full((B,), nf, dtype=torch.int32, device=dev)
    node_stride = nf * npf
    d_nodes = torch.empty(B, node_stride * 8, dtype=torch.uint8, device=dev)
    d_nn = torch.zeros(B, dtype=torch.int32, device=dev)
    RS = 16 if os.environ.get('DEC_DBG') else 8
    d_rst = torch.zeros(B, RS, dtype=torch.int32, device=dev)
    d_nr = torch.zeros(B, dtype=torch.int32, device=dev)
    FRAMED = os.environ.get('DEC_FRAMED')  # the same frames, their offsets (k * S) and zero gap flags given
    if FRAMED:
        d_off = (torch.arange(nf, dtype=torch.int32, device=dev) * S).repeat(B, 1).contiguous()
        d_gap = torch.zeros(B, nf, dtype=torch.uint8, device=dev)
    def run():
        gpu.decode_batch_dev(ans, 125, buf.data_ptr(), nf * S, d_off.data_ptr() if FRAMED else 0,
                             d_gap.data_ptr() if FRAMED else 0, d_nf.data_ptr(), nf, B, 0, 0,
                             d_nodes.data_ptr(), node_stride, d_nn.data_ptr(), d_rst.data_ptr(), RS, d_nr.data_ptr())
    PIECES = int(os.environ.get('DEC_PIECES', '0'))
    if PIECES > 1:
        # timing experiment: the call as PIECES calls of nf / PIECES frames (+ the overlap frame), each short
        # enough for the staged decoder; outputs of the later pieces go to the same buffer (contents unused)
        W = (nf + PIECES - 1) // PIECES
        d_nfp = [torch.full((B,), min(W + (1 if p else 0), nf - p * W + (1 if p else 0)), dtype=torch.int32, device=dev) for p in range(PIECES)]
        d_state = torch.zeros(B, 4, dtype=torch.int32, device=dev)
        d_state2 = torch.zeros(B, 4, dtype=torch.int32, device=dev); d_state2[:, 2] = 1
        def run():
            for p in range(PIECES):
                first = p * W - (1 if p else 0)
                gpu.decode_batch_dev(ans, 125, buf.data_ptr() + first * S, nf * S, 0, 0, d_nfp[p].data_ptr(), W + 1, B,
                                     d_state2.data_ptr() if p else d_state.data_ptr(), 0,
                                     d_nodes.data_ptr(), node_stride, d_nn.data_ptr(), d_rst.data_ptr(), RS, d_nr.data_ptr())
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); run(); b.record(st); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    nodes = int(d_nn.sum().item())
    t = min(ts)
    if os.environ.get('DEC_SUM'):
        w = d_nodes.view(torch.int32).to(torch.int64)
        k = (torch.arange(w.shape[1], device=dev) % 1021 + 1)
        print("   checksum", int(w.sum().item()), int((w * k).sum().item()), "resets", int(d_nr.sum().item()))
    if os.environ.get('DEC_DBG'):
        ph = d_rst.cpu().numpy()[:, 3:8].astype(np.float64)
        print("   phase cycles P1 P2 P3 P4 P5 (mean):", ph.mean(0).round(0), "sum", ph.sum(1).mean().round(0), "head (staging)", d_rst.cpu().numpy()[:, 2].astype(np.float64).mean().round(0))
        if ans == 0x86:
            print("   ultra-dense smoothing: [walk, map scan, second walk, -] (mean cycles):", d_rst.cpu().numpy()[:, 8:12].astype(np.float64).mean(0).round(0))
        print("   P1 p10/p50/p90:", np.percentile(ph[:, 0], [10, 50, 90]).round(0), " first 1792 WGs mean", ph[:1792, 0].mean().round(0), "rest", ph[1792:, 0].mean().round(0),
              " P3 p10/p50/p90:", np.percentile(ph[:, 2], [10, 50, 90]).round(0))
    print(f"ans {ans:#x}: {nodes/1e6:.1f} Mnodes in {t:.3f} ms -> {nodes/t/1e6:.1f} Gnodes/s, "
          f"in {B*nf*S/1e6:.0f} MB out {nodes*8/1e6:.0f} MB -> {(B*nf*S+nodes*8)/t/1e6:.0f} GB/s")
    if True:
        scan_cap, n_stride = 4, min(node_stride, 32768)
        d_batch = torch.empty(B * scan_cap, n_stride * 8, dtype=torch.uint8, device=dev)
        d_len = torch.zeros(B * scan_cap, dtype=torch.int32, device=dev)
        d_ns2 = torch.zeros(B, dtype=torch.int32, device=dev)
        d_st2 = torch.zeros(B, dtype=torch.int32, device=dev)
        def fused():
            gpu.decode_scans_dev(ans, 125, buf.data_ptr(), nf * S, d_off.data_ptr() if FRAMED else 0,
                                 d_gap.data_ptr() if FRAMED else 0, d_nf.data_ptr(), nf, B, 0, 0, 32768,
                                 d_batch.data_ptr(), n_stride, scan_cap, d_len.data_ptr(), d_ns2.data_ptr(), 0, d_st2.data_ptr())
        fused(); torch.cuda.synchronize(); ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st); fused(); b.record(st); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        extra = ""
        if os.environ.get('DEC_SUM'):
            d_batch.zero_(); fused(); torch.cuda.synchronize()
            w = d_batch.view(torch.int32).to(torch.int64)
            k = (torch.arange(w.shape[1], device=dev) % 1021 + 1)
            extra = f", checksum {int(w.sum().item())} {int((w * k).sum().item())}"
            del w
        print(f"  segment-fused decode_scans_dev: {min(ts):.3f} ms, scans {int(d_ns2.sum().item())}, nodes in scans {int(d_len.sum().item())}, status {int(d_st2.max().item())}" + extra)
        del d_batch
    if ans == 0x85:
        d_seg = torch.empty_like(d_nodes)
        scan_cap = 8
        d_off = torch.zeros(B, scan_cap + 1, dtype=torch.int32, device=dev)
        d_ns = torch.zeros(B, dtype=torch.int32, device=dev)
        def seg():
            gpu.segment_batch_dev(d_nodes.data_ptr(), node_stride, d_nn.data_ptr(), d_rst.data_ptr(), 8,
                                  d_nr.data_ptr(), B, 32768, d_seg.data_ptr(), node_stride, d_off.data_ptr(), scan_cap, d_ns.data_ptr())
        seg(); torch.cuda.synchronize(); ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st); seg(); b.record(st); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        print(f"  segment: {min(ts):.3f} ms, scans {int(d_ns.sum().item())}")
    del buf, d_nodes
