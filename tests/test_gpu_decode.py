"""GPU parity tests of the decode stage: recorded answer streams ->
nodes (k_decode) -> completed scans (k_segment) -> scan batch, through the C ABI, against the
CPU oracle (oracle/oracle_unpack.cpp, itself pinned against the genuine SDK) and against the
golden vectors produced by the genuine SDK unpackers.  Integer work: bit-exact everywhere."""
from pathlib import Path

import numpy as np
import pytest

from rplidar_ros2_driver_amd import NODE_DTYPE, Params, abi
from rplidar_ros2_driver_amd import capsules as cp
from tests import oracle_lib

pytestmark = pytest.mark.gpu

GOLD = Path(__file__).resolve().parent / "golden" / "unpack_golden.npz"
ALL_ANS = (0x81, 0x82, 0x83, 0x84, 0x85, 0x86)


def _torch():
    import torch
    return torch


def _check_stream(gpu, oracle, ans, data, dur, state=(0, 0)):
    nodes, rst, err, st = gpu.decode_stream(ans, data, dur, state)
    w_nodes, w_rst, w_err, w_st = oracle.unpack(ans, data, dur, state=state)
    assert len(nodes) == len(w_nodes)
    if nodes.tobytes() != w_nodes.tobytes():
        d = np.nonzero(nodes != w_nodes)[0]
        raise AssertionError(f"ans {ans:#x}: {len(d)} nodes differ, first at {d[:4]}: "
                             f"{nodes[d[:2]]} != {w_nodes[d[:2]]}")
    assert list(rst) == list(w_rst)
    assert err == w_err
    assert st == w_st
    return nodes, rst, st


def test_decode_matches_genuine_golden(gpu):
    """The kernels against what the GENUINE SDK unpackers published (tests/golden)."""
    g = np.load(GOLD)
    tags = sorted({k.split("__")[0] for k in g.files})
    for t in tags:
        ans, nf, seed, corrupt, dur, dense_last = (int(v) for v in g[t + "__meta"])
        nodes, rst, err, _ = gpu.decode_stream(ans, g[t + "__bytes"], dur, (dense_last, 0))
        assert nodes.tobytes() == g[t + "__nodes"].tobytes(), t
        assert list(rst) == list(g[t + "__reset_at"]), t
        assert err == int(g[t + "__n_err"]), t


@pytest.mark.parametrize("ans", ALL_ANS)
@pytest.mark.parametrize("corrupt", [False, True])
def test_decode_stream_matches_oracle(gpu, oracle, ans, corrupt):
    for seed in range(6):
        fpr = [12.3, 3.1, 40.0, 7.7, 300.0, 1.5][seed]
        dur = [125, 125, 32, 20, 2, 1000000][seed]
        payload = "random" if seed % 2 else "ring"
        nf = 300 if ans != 0x81 else 3000
        data = cp.make_stream(ans, nf, 40 + seed, corrupt=corrupt, payload=payload,
                              frames_per_rev=fpr)
        _check_stream(gpu, oracle, ans, data, dur, state=(seed & 1, 0))


@pytest.mark.parametrize("ans", [0x85, 0x86])
def test_decode_sync_filter_runs(gpu, oracle, ans):
    """Long runs of raw sync bits (tiny angle steps around 0 deg, start angles above 360 deg)
    and the carried-in sync bit: s_i = r_i & ~s_{i-1}."""
    S = cp.FRAME_SIZE[ans]
    o = 2 if ans == 0x85 else 8
    for start_q6, step_q6 in ((0, 0), (0, 1), (23039, 1), (23030, 3), (32000, 5), (23040, 0)):
        f = cp.make_frames(ans, 40, 9, payload="random", first_sync=False)
        sa = ((start_q6 + step_q6 * np.arange(40)) % 32768).astype(np.uint16)
        f[:, o] = sa & 0xFF
        f[:, o + 1] = sa >> 8
        cp._seal_capsules(f)
        for init in (0, 1):
            nodes, _, _ = _check_stream(gpu, oracle, ans, f.reshape(-1), 125, state=(init, 0))
            assert len(nodes) == 39 * cp.NODES_PER_FRAME[ans]
    assert S in (84, 170)


def test_ultra_dense_smoothing_chains(gpu, oracle):
    """Scale-0 distances built to smooth over long chains, break chains at every distance,
    hit zero, and cross capsule boundaries (handler_capsules.cpp:997-1003)."""
    # (400 frames = 25 600 nodes: the kernel smooths in chunks of 8192 nodes, chains cross them)
    for nfr in (60, 400):
        rng = np.random.default_rng(5 + nfr)
        f = cp.make_frames(0x86, nfr, 3, payload="random", first_sync=False)
        walk = np.cumsum(rng.integers(-1, 2, nfr * 64)) + 500  # dist_q2 = 8 * walk: steps of 0 / 8
        walk[rng.random(nfr * 64) < 0.05] += 40          # chain breaks
        walk[rng.random(nfr * 64) < 0.03] = 0            # zeros
        walk = np.clip(walk, 0, 0xFFC // 4).astype(np.uint32)
        scale = np.where(rng.random(nfr * 64) < 0.1, rng.integers(1, 4, nfr * 64), 0).astype(np.uint32)
        w = ((walk << 2) & 0xFFC) | scale | (rng.integers(0, 256, nfr * 64).astype(np.uint32) << 12)
        w = w.reshape(nfr, 64)
        e, o_ = w[:, 0::2], w[:, 1::2]
        f[:, 10::5] = e & 0xFF
        f[:, 11::5] = (e >> 8) & 0xFF
        f[:, 12::5] = o_ & 0xFF
        f[:, 13::5] = (o_ >> 8) & 0xFF
        f[:, 14::5] = ((e >> 16) & 0xF) | (((o_ >> 16) & 0xF) << 4)
        cp._seal_capsules(f)
        for last in (0, 5, 8000, 8190):
            _check_stream(gpu, oracle, 0x86, f.reshape(-1), 125, state=(0, last))
        # a constant distance entered from 8 below: the smoothed value settles one unit under the
        # raw one and stays there ((x >> 1) keeps -1), so whole thread segments are entered in a
        # state other than "not smoothed" and never forget it (the kernel's guess is wrong for
        # every one of them); a zero now and then restarts the chain from the other fixed point
        walk = np.full(nfr * 64, 700, np.uint32)
        walk[0] = 0
        walk[1] = 699
        z = np.nonzero(rng.random(nfr * 64) < 0.002)[0]
        z = z[(z > 2) & (z < nfr * 64 - 2)]
        walk[z] = 0
        walk[z[::2] + 1] = 699
        w = ((walk << 2) & 0xFFC) | (rng.integers(0, 256, nfr * 64).astype(np.uint32) << 12)
        w = w.reshape(nfr, 64)
        e, o_ = w[:, 0::2], w[:, 1::2]
        f[:, 10::5] = e & 0xFF
        f[:, 11::5] = (e >> 8) & 0xFF
        f[:, 12::5] = o_ & 0xFF
        f[:, 13::5] = (o_ >> 8) & 0xFF
        f[:, 14::5] = ((e >> 16) & 0xF) | (((o_ >> 16) & 0xF) << 4)
        cp._seal_capsules(f)
        for last in (0, 5599, 5600):
            nodes, _, _ = _check_stream(gpu, oracle, 0x86, f.reshape(-1), 125, state=(0, last))
        d = nodes["dist_mm_q2"]
        assert (d == 5599).sum() > len(d) // 4  # (the -1 fixed point really is where it sits)


def test_decode_state_carries_between_calls(gpu, oracle):
    """Two calls on the halves of one dense / ultra-dense stream == the oracle run the same way
    (the latch restarts at a call boundary; the sync / smoothing state is carried)."""
    for ans in (0x85, 0x86):
        S = cp.FRAME_SIZE[ans]
        data = cp.make_stream(ans, 120, 77, payload="ring", frames_per_rev=9.7)
        a, b_ = data[: 60 * S], data[60 * S:]
        _, _, st = _check_stream(gpu, oracle, ans, a, 125, state=(0, 0))
        _check_stream(gpu, oracle, ans, b_, 125, state=st)


@pytest.mark.parametrize("ans,nf", [(0x85, 5000), (0x86, 1300), (0x82, 4100), (0x84, 2100),
                                    (0x83, 2100), (0x81, 20000)])
def test_decode_long_recording_is_cut_without_loss(gpu, oracle, ans, nf):
    """A recording longer than one decode call holds: rplgpu_decode_stream cuts it into pieces
    that overlap by one frame (state flags bit 0) — same nodes, resets and error count as the
    oracle's single pass, corrupted stream included."""
    for corrupt in (False, True):
        data = cp.make_stream(ans, nf, 31, corrupt=corrupt, payload="ring", frames_per_rev=33.3)
        _check_stream(gpu, oracle, ans, data, 125, state=(0, 0))


def test_decode_batch_dev_unframed_and_status(gpu, oracle):
    """Device-resident batch: frames back to back (no offsets); a stream whose frames do not all
    start with the sync pattern is flagged RPLGPU_STREAM_UNFRAMED and yields nothing."""
    torch = _torch()
    dev = torch.device("cuda:0")
    for ans in (0x82, 0x84, 0x85, 0x86, 0x83, 0x81):
        S, npf = cp.FRAME_SIZE[ans], cp.NODES_PER_FRAME[ans]
        B, nf = 7, 90
        streams = [cp.make_stream(ans, nf - 3 * b, 200 + b, payload=("random", "ring", "ring_near")[b % 3],
                                  frames_per_rev=11.0 + b) for b in range(B)]
        streams[4] = streams[4].copy()
        streams[4][5 * S] ^= 0xF0 if ans != 0x81 else 0x01  # broken sync pattern in frame 5
        stride = nf * S
        buf = np.zeros((B, stride), np.uint8)
        for b, s in enumerate(streams):
            buf[b, : len(s)] = s
        nfs = np.array([len(s) // S for s in streams], np.int32)
        d_bytes = torch.from_numpy(buf).to(dev)
        d_nf = torch.from_numpy(nfs).to(dev)
        node_stride = nf * npf
        d_nodes = torch.zeros(B, node_stride * 8, dtype=torch.uint8, device=dev)
        d_nn = torch.zeros(B, dtype=torch.int32, device=dev)
        d_rst = torch.zeros(B, 16, dtype=torch.int32, device=dev)
        d_nr = torch.zeros(B, dtype=torch.int32, device=dev)
        d_ne = torch.zeros(B, dtype=torch.int32, device=dev)
        d_st = torch.zeros(B, dtype=torch.int32, device=dev)
        d_state = torch.zeros(B, 4, dtype=torch.int32, device=dev)
        gpu.decode_batch_dev(ans, 125, d_bytes.data_ptr(), stride, 0, 0, d_nf.data_ptr(), nf, B,
                             0, d_state.data_ptr(), d_nodes.data_ptr(), node_stride,
                             d_nn.data_ptr(), d_rst.data_ptr(), 16, d_nr.data_ptr(),
                             d_ne.data_ptr(), d_st.data_ptr())
        gpu.synchronize()
        nn, st = d_nn.cpu().numpy(), d_st.cpu().numpy()
        nodes = d_nodes.cpu().numpy().view(NODE_DTYPE).reshape(B, node_stride)
        rst, nr = d_rst.cpu().numpy(), d_nr.cpu().numpy()
        state = d_state.cpu().numpy()
        for b in range(B):
            if b == 4:
                assert st[b] & abi_status("UNFRAMED") and nn[b] == 0
                continue
            want, w_rst, w_err, w_st = oracle.unpack(ans, streams[b], 125)
            assert st[b] == 0 and nn[b] == len(want), (hex(ans), b)
            assert nodes[b, : nn[b]].tobytes() == want.tobytes(), (hex(ans), b)
            assert list(rst[b, : nr[b]]) == list(w_rst)
            assert tuple(state[b, :2]) == w_st


def abi_status(name):
    return {"UNFRAMED": 0x10, "FRAMES_TRUNCATED": 0x20, "RESETS_TRUNCATED": 0x40}[name]


def test_segment_matches_oracle_and_golden(gpu, oracle):
    torch = _torch()
    dev = torch.device("cuda:0")
    g = np.load(GOLD)
    tags = sorted({k.split("__")[0] for k in g.files})
    cases = [(g[t + "__nodes"], g[t + "__reset_at"]) for t in tags]
    # plus long synthetic node streams with many revolutions and resets
    rng = np.random.default_rng(3)
    for _ in range(4):
        n = 20000
        nd = np.zeros(n, NODE_DTYPE)
        nd["dist_mm_q2"] = rng.integers(0, 1 << 20, n)
        nd["angle_z_q14"] = rng.integers(0, 65536, n)
        sync = rng.random(n) < 0.004
        nd["flag"] = np.where(sync, 1, 2)
        rs = np.unique(rng.integers(0, n + 1, 12)).astype(np.uint32)
        cases.append((nd, rs))
    B = len(cases)
    node_stride = max(len(c[0]) for c in cases) + 8
    rcap = max(len(c[1]) for c in cases) + 1
    for max_count in (8192, 37, 1):
        buf = np.zeros((B, node_stride), NODE_DTYPE)
        rbuf = np.zeros((B, rcap), np.uint32)
        nn = np.zeros(B, np.int32)
        nr = np.zeros(B, np.int32)
        for b, (nd, rs) in enumerate(cases):
            buf[b, : len(nd)] = nd
            rbuf[b, : len(rs)] = rs
            nn[b], nr[b] = len(nd), len(rs)
        scan_cap = 4096
        d_nodes = torch.from_numpy(buf.view(np.uint8).reshape(B, -1)).to(dev)
        d_out = torch.zeros_like(d_nodes)
        d_nn, d_nr = torch.from_numpy(nn).to(dev), torch.from_numpy(nr).to(dev)
        d_rst = torch.from_numpy(rbuf.view(np.int32)).to(dev)
        d_off = torch.zeros(B, scan_cap + 1, dtype=torch.int32, device=dev)
        d_ns = torch.zeros(B, dtype=torch.int32, device=dev)
        d_st = torch.zeros(B, dtype=torch.int32, device=dev)
        gpu.segment_batch_dev(d_nodes.data_ptr(), node_stride, d_nn.data_ptr(), d_rst.data_ptr(),
                              rcap, d_nr.data_ptr(), B, max_count, d_out.data_ptr(), node_stride,
                              d_off.data_ptr(), scan_cap, d_ns.data_ptr(), d_st.data_ptr())
        gpu.synchronize()
        out = d_out.cpu().numpy().view(NODE_DTYPE).reshape(B, node_stride)
        offs, ns = d_off.cpu().numpy(), d_ns.cpu().numpy()
        assert np.all(d_st.cpu().numpy() == 0)
        for b, (nd, rs) in enumerate(cases):
            want, w_off = oracle.segment(nd, rs, max_count)
            assert ns[b] == len(w_off) - 1, (b, max_count)
            assert list(offs[b, : ns[b] + 1]) == list(w_off), (b, max_count)
            assert out[b, : w_off[-1]].tobytes() == want.tobytes(), (b, max_count)


def test_recorded_stream_to_laserscan_pipeline(gpu, oracle):
    """Config-2-like end to end on the device: DenseBoost capsule streams -> decode -> scan
    assembly -> scan batch -> ascendScanData -> publish_scan, equal to the oracle chain."""
    torch = _torch()
    dev = torch.device("cuda:0")
    ans, S, npf = 0x85, 84, 40
    B, nf = 5, 400
    streams = [cp.make_stream(ans, nf, 900 + b, payload="ring", frames_per_rev=20.0 + 3 * b)
               for b in range(B)]
    buf = np.stack(streams)
    d_bytes = torch.from_numpy(buf).to(dev)
    d_nf = torch.full((B,), nf, dtype=torch.int32, device=dev)
    node_stride = nf * npf
    d_nodes = torch.zeros(B, node_stride * 8, dtype=torch.uint8, device=dev)
    d_seg = torch.zeros_like(d_nodes)
    d_nn = torch.zeros(B, dtype=torch.int32, device=dev)
    d_rst = torch.zeros(B, 8, dtype=torch.int32, device=dev)
    d_nr = torch.zeros(B, dtype=torch.int32, device=dev)
    scan_cap, n_stride, max_scans = 64, 2048, 256
    d_off = torch.zeros(B, scan_cap + 1, dtype=torch.int32, device=dev)
    d_ns = torch.zeros(B, dtype=torch.int32, device=dev)
    d_base = torch.zeros(B + 1, dtype=torch.int32, device=dev)
    d_batch = torch.zeros(max_scans, n_stride * 8, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(max_scans, dtype=torch.int32, device=dev)
    gpu.decode_batch_dev(ans, 125, d_bytes.data_ptr(), nf * S, 0, 0, d_nf.data_ptr(), nf, B, 0, 0,
                         d_nodes.data_ptr(), node_stride, d_nn.data_ptr(), d_rst.data_ptr(), 8,
                         d_nr.data_ptr())
    gpu.segment_batch_dev(d_nodes.data_ptr(), node_stride, d_nn.data_ptr(), d_rst.data_ptr(), 8,
                          d_nr.data_ptr(), B, 8192, d_seg.data_ptr(), node_stride,
                          d_off.data_ptr(), scan_cap, d_ns.data_ptr())
    gpu.scans_to_batch_dev(d_seg.data_ptr(), node_stride, d_off.data_ptr(), scan_cap,
                           d_ns.data_ptr(), B, d_base.data_ptr(), d_batch.data_ptr(), n_stride,
                           max_scans, d_len.data_ptr())
    gpu.synchronize()
    total = int(d_base.cpu().numpy()[B])
    assert 0 < total <= max_scans
    d_status = torch.zeros(max_scans, dtype=torch.int32, device=dev)
    gpu.ascend_batch_dev(d_batch.data_ptr(), n_stride, d_len.data_ptr(), total, d_status.data_ptr())
    p = Params.defaults(range_max=40.0)
    d_r = torch.zeros(total, n_stride, dtype=torch.float32, device=dev)
    d_i = torch.zeros(total, n_stride, dtype=torch.float32, device=dev)
    d_cnt = torch.zeros(total, dtype=torch.int32, device=dev)
    gpu.laserscan_batch_dev(d_batch.data_ptr(), n_stride, d_len.data_ptr(), total, p,
                            d_r.data_ptr(), d_i.data_ptr(), d_cnt.data_ptr())
    gpu.synchronize()
    lens = d_len.cpu().numpy()
    r, i, cnt = d_r.cpu().numpy(), d_i.cpu().numpy(), d_cnt.cpu().numpy()
    gidx = 0
    for b in range(B):
        nodes, rst, _, _ = oracle.unpack(ans, streams[b], 125)
        scans, offs = oracle.segment(nodes, rst, 8192)
        for s in range(len(offs) - 1):
            scan = scans[offs[s]: offs[s + 1]]
            assert lens[gidx] == len(scan)
            asc, res = oracle.ascend(scan)
            assert res == 0
            wr, wi, wm = oracle.publish_scan(asc, oracle_lib.copy_params(p), 0.1)
            assert cnt[gidx] == wm.count
            assert r[gidx, : wm.count].tobytes() == wr.tobytes()
            # equal (angle, dist) ties may carry either quality (documented tie rule)
            assert np.count_nonzero(i[gidx, : wm.count] != wi) <= wm.count // 50
            gidx += 1
    assert gidx == total


@pytest.mark.parametrize("ans", ALL_ANS)
def test_decode_scans_matches_oracle_chain(gpu, oracle, ans):
    """rplgpu_decode_scans_dev (decode + scan assembly straight into batch slots, the decoder's
    own sync list) against oracle.unpack -> oracle.segment, stream by stream: clean and corrupted
    streams, framed (offsets + gaps after junk bytes) and back to back, several max_count /
    n_stride / scan_cap combinations (truncation flags), and the slots a stream leaves empty."""
    torch = _torch()
    dev = torch.device("cuda:0")
    S, npf = cp.FRAME_SIZE[ans], cp.NODES_PER_FRAME[ans]
    nf = 220 if ans != 0x81 else 6000
    rng = np.random.default_rng(ans)
    for framed in (False, True):
        streams = []
        for b in range(9):
            fpr = [9.3, 25.0, 4.2, 60.0, 2.5, 14.0, 33.3, 7.0, 110.0][b]
            if ans == 0x81:
                fpr *= 40
            d = cp.make_stream(ans, nf - 7 * b, 500 + b, corrupt=framed and b % 3 == 1,
                               payload=("random", "ring", "ring_near")[b % 3], frames_per_rev=fpr)
            if framed and b % 2 == 0:  # junk between frames: gaps clear the capsule latch
                cut = (len(d) // S // 2) * S
                d = np.concatenate([d[:cut], rng.integers(0, 256, 13, dtype=np.uint8), d[cut:]])
            streams.append(d)
        B = len(streams)
        stride = max(len(d) for d in streams)
        buf = np.zeros((B, stride), np.uint8)
        offs = np.zeros((B, nf), np.uint32)
        gaps = np.zeros((B, nf), np.uint8)
        nfs = np.zeros(B, np.int32)
        for b, d in enumerate(streams):
            buf[b, : len(d)] = d
            if framed:
                o, g = abi.frame_stream(ans, d)
                nfs[b] = len(o)
                offs[b, : len(o)], gaps[b, : len(o)] = o, g
            else:
                nfs[b] = len(d) // S
        d_bytes = torch.from_numpy(buf).to(dev)
        d_off = torch.from_numpy(offs.view(np.int32)).to(dev)
        d_gap = torch.from_numpy(gaps).to(dev)
        d_nf = torch.from_numpy(nfs).to(dev)
        want = []
        for d in streams:
            nodes, rst, err, _ = oracle.unpack(ans, d, 125)
            want.append((nodes, rst, err))
        for max_count, n_stride, scan_cap in ((8192, 4096, 64), (37, 64, 64), (8192, 50, 3), (1, 8, 512)):
            d_batch = torch.zeros(B * scan_cap, n_stride * 8, dtype=torch.uint8, device=dev)
            d_len = torch.full((B * scan_cap,), -1, dtype=torch.int32, device=dev)
            d_ns = torch.zeros(B, dtype=torch.int32, device=dev)
            d_ne = torch.zeros(B, dtype=torch.int32, device=dev)
            d_st = torch.zeros(B, dtype=torch.int32, device=dev)
            gpu.decode_scans_dev(ans, 125, d_bytes.data_ptr(), stride,
                                 d_off.data_ptr() if framed else 0, d_gap.data_ptr() if framed else 0,
                                 d_nf.data_ptr(), nf, B, 0, 0, max_count, d_batch.data_ptr(), n_stride,
                                 scan_cap, d_len.data_ptr(), d_ns.data_ptr(), d_ne.data_ptr(),
                                 d_st.data_ptr())
            gpu.synchronize()
            batch = d_batch.cpu().numpy().view(NODE_DTYPE).reshape(B * scan_cap, n_stride)
            lens, ns, ne, st = (t.cpu().numpy() for t in (d_len, d_ns, d_ne, d_st))
            for b in range(B):
                nodes, rst, err = want[b]
                scans, w_off = oracle.segment(nodes, rst, max_count)
                n_want = len(w_off) - 1
                key = (hex(ans), framed, b, max_count, n_stride, scan_cap)
                assert ne[b] == err, key
                assert ns[b] == min(n_want, scan_cap), key
                assert bool(st[b] & abi_status("RESETS_TRUNCATED")) == (n_want > scan_cap), key
                too_long = False
                for s_ in range(scan_cap):
                    g = b * scan_cap + s_
                    if s_ >= n_want:
                        assert lens[g] == 0, key
                        continue
                    scan = scans[w_off[s_]: w_off[s_ + 1]]
                    keep = min(len(scan), n_stride)
                    too_long |= len(scan) > n_stride
                    assert lens[g] == keep, key
                    assert batch[g, :keep].tobytes() == scan[:keep].tobytes(), key
                assert bool(st[b] & 0x8) == too_long, key  # RPLGPU_SCAN_OUT_TRUNCATED
    # the slot layout feeds the batch entry points as it is: empty slots are empty scans
    p = Params.defaults(range_max=40.0)
    total = min(B * scan_cap, 4096)
    lens = lens[:total]
    d_status = torch.zeros(total, dtype=torch.int32, device=dev)
    gpu.ascend_batch_dev(d_batch.data_ptr(), n_stride, d_len.data_ptr(), total, d_status.data_ptr())
    d_r = torch.zeros(total, n_stride, dtype=torch.float32, device=dev)
    d_i = torch.zeros(total, n_stride, dtype=torch.float32, device=dev)
    d_cnt = torch.full((total,), -1, dtype=torch.int32, device=dev)
    gpu.laserscan_batch_dev(d_batch.data_ptr(), n_stride, d_len.data_ptr(), total, p,
                            d_r.data_ptr(), d_i.data_ptr(), d_cnt.data_ptr())
    gpu.synchronize()
    cnt = d_cnt.cpu().numpy()
    assert np.all(cnt[lens == 0] == 0)


@pytest.mark.parametrize("ans", [0x82, 0x84, 0x85, 0x86])
def test_decode_scans_streams_beyond_the_fused_tables(gpu, oracle, ans):
    """The fused decoder of the express / ultra / dense types holds 256 sync nodes and 64 reset
    requests per stream; streams beyond that (a revolution per capsule, corrupted headers) must
    come out the same through the general path, next to streams the fused path handles."""
    torch = _torch()
    dev = torch.device("cuda:0")
    S = cp.FRAME_SIZE[ans]
    # (ultra-dense: a call takes at most 512 frames and the decoder rejects capsules that sweep
    # more than ~1/3 of a turn, so its streams stay INSIDE the fused tables — 256 sync nodes, 64
    # resets — whatever the recording: they are here for the comparison, not for the fallback)
    nf = 1200 if ans != 0x86 else 500
    fprs = (3.0, 30.0, 2.2, 3.7, 55.0, 2.6) if ans != 0x86 else (3.0, 30.0, 2.6, 3.7, 55.0, 2.7)
    streams = [cp.make_stream(ans, nf, 900 + b, corrupt=(b == 2), payload="random" if b % 2 else "ring",
                              frames_per_rev=fpr) for b, fpr in enumerate(fprs)]
    B = len(streams)
    stride = max(len(d) for d in streams)
    mf = max(len(d) for d in streams) // S + 1
    buf = np.zeros((B, stride), np.uint8)
    offs = np.zeros((B, mf), np.uint32)
    gaps = np.zeros((B, mf), np.uint8)
    nfs = np.zeros(B, np.int32)
    for b, d in enumerate(streams):  # (host framing: the corrupted stream has junk between frames)
        buf[b, : len(d)] = d
        o, g = abi.frame_stream(ans, d)
        nfs[b] = len(o)
        offs[b, : len(o)], gaps[b, : len(o)] = o, g
    d_bytes = torch.from_numpy(buf).to(dev)
    d_off = torch.from_numpy(offs.view(np.int32)).to(dev)
    d_gap = torch.from_numpy(gaps).to(dev)
    d_nf = torch.from_numpy(nfs).to(dev)
    scan_cap, n_stride = 512, 256
    d_batch = torch.zeros(B * scan_cap, n_stride * 8, dtype=torch.uint8, device=dev)
    d_len = torch.full((B * scan_cap,), -1, dtype=torch.int32, device=dev)
    d_ns = torch.zeros(B, dtype=torch.int32, device=dev)
    d_ne = torch.zeros(B, dtype=torch.int32, device=dev)
    d_st = torch.zeros(B, dtype=torch.int32, device=dev)
    gpu.decode_scans_dev(ans, 125, d_bytes.data_ptr(), stride, d_off.data_ptr(), d_gap.data_ptr(),
                         d_nf.data_ptr(), mf, B, 0, 0, 8192, d_batch.data_ptr(), n_stride, scan_cap, d_len.data_ptr(), d_ns.data_ptr(),
                         d_ne.data_ptr(), d_st.data_ptr())
    gpu.synchronize()
    batch = d_batch.cpu().numpy().view(NODE_DTYPE).reshape(B * scan_cap, n_stride)
    lens, ns, ne = d_len.cpu().numpy(), d_ns.cpu().numpy(), d_ne.cpu().numpy()
    many = 0
    for b, d in enumerate(streams):
        nodes, rst, err, _ = oracle.unpack(ans, d, 125)
        scans, w_off = oracle.segment(nodes, rst, 8192)
        n_want = len(w_off) - 1
        many += n_want > 256 or len(rst) > 64
        assert ne[b] == err and ns[b] == min(n_want, scan_cap), (hex(ans), b)
        for s_ in range(scan_cap):
            g = b * scan_cap + s_
            if s_ >= n_want:
                assert lens[g] == 0
                continue
            scan = scans[w_off[s_]: w_off[s_ + 1]]
            keep = min(len(scan), n_stride)
            assert lens[g] == keep and batch[g, :keep].tobytes() == scan[:keep].tobytes(), (hex(ans), b, s_)
    assert many >= 2 or ans == 0x86  # the general path really was taken


@pytest.mark.parametrize("ans", [0x82, 0x84, 0x85, 0x86])
def test_decode_scans_carried_state_through_the_fused_path(gpu, oracle, ans):
    """rplgpu_decode_scans_dev with a non-null state_in / state_out: the second halves of the
    streams start from the state the first halves left (sync filter of the dense types, the
    ultra-dense smoothing distance), exactly as oracle.unpack run the same way; the scans of each
    half are oracle.segment of that half's nodes (a scan across the cut belongs to neither call)."""
    torch = _torch()
    dev = torch.device("cuda:0")
    S = cp.FRAME_SIZE[ans]
    nf, B = 240, 5
    streams = [cp.make_stream(ans, nf, 4100 + b, payload=("random", "ring", "ring_near")[b % 3],
                              frames_per_rev=(8.3, 21.0, 5.1, 40.0, 12.9)[b]) for b in range(B)]
    cut = (nf // 2) * S
    halves = [[d[:cut] for d in streams], [d[cut:] for d in streams]]
    scan_cap, n_stride = 64, 2048
    states = np.zeros((B, 4), np.int32)
    want_state = [(0, 0)] * B
    for part in halves:
        stride = max(len(d) for d in part)
        buf = np.zeros((B, stride), np.uint8)
        nfs = np.zeros(B, np.int32)
        for b, d in enumerate(part):
            buf[b, : len(d)] = d
            nfs[b] = len(d) // S
        d_bytes = torch.from_numpy(buf).to(dev)
        d_nf = torch.from_numpy(nfs).to(dev)
        d_sin = torch.from_numpy(states.copy()).to(dev)
        d_sout = torch.zeros(B, 4, dtype=torch.int32, device=dev)
        d_batch = torch.zeros(B * scan_cap, n_stride * 8, dtype=torch.uint8, device=dev)
        d_len = torch.full((B * scan_cap,), -1, dtype=torch.int32, device=dev)
        d_ns = torch.zeros(B, dtype=torch.int32, device=dev)
        d_ne = torch.zeros(B, dtype=torch.int32, device=dev)
        d_st = torch.zeros(B, dtype=torch.int32, device=dev)
        gpu.decode_scans_dev(ans, 125, d_bytes.data_ptr(), stride, 0, 0, d_nf.data_ptr(), int(nfs.max()), B,
                             d_sin.data_ptr(), d_sout.data_ptr(), 8192, d_batch.data_ptr(), n_stride,
                             scan_cap, d_len.data_ptr(), d_ns.data_ptr(), d_ne.data_ptr(), d_st.data_ptr())
        gpu.synchronize()
        batch = d_batch.cpu().numpy().view(NODE_DTYPE).reshape(B * scan_cap, n_stride)
        lens, ns = d_len.cpu().numpy(), d_ns.cpu().numpy()
        states = d_sout.cpu().numpy()
        for b, d in enumerate(part):
            nodes, rst, err, st_out = oracle.unpack(ans, d, 125, state=want_state[b])
            want_state[b] = st_out
            scans, w_off = oracle.segment(nodes, rst, 8192)
            assert ns[b] == len(w_off) - 1 and int(d_ne[b]) == err, (hex(ans), b)
            assert (int(states[b, 0]), int(states[b, 1])) == st_out, (hex(ans), b)
            for s_ in range(ns[b]):
                want = scans[w_off[s_]: w_off[s_ + 1]]
                keep = min(len(want), n_stride)  # (a batch slot holds n_stride nodes)
                assert lens[b * scan_cap + s_] == keep
                assert batch[b * scan_cap + s_, :keep].tobytes() == want[:keep].tobytes(), (hex(ans), b, s_)


@pytest.mark.parametrize("ans", ALL_ANS)
def test_long_recording_becomes_scans_across_calls(gpu, oracle, ans):
    """rplgpu_decode_scans_carry_dev: recordings cut into pieces of a few dozen frames (cuts anywhere:
    inside scans, right at sync nodes, pieces without any sync node, pieces with corrupted frames that
    raise scan-reset requests), decoder state and the open scan carried from call to call.  All
    pieces together must deliver exactly the scans ONE pass of the unpacker + ScanDataHolder makes of
    the whole recording (oracle.unpack + oracle.segment, pinned against the genuine SDK), in order."""
    torch = _torch()
    dev = torch.device("cuda:0")
    S = cp.FRAME_SIZE[ans]
    # (HQ frames carry their 96 sync flags in a random payload: dozens of tiny scans per frame, so
    # fewer frames, more and shorter batch slots)
    B, nf = 6, (420 if ans != 0x83 else 40)
    rng = np.random.default_rng(ans)
    streams = []
    for b in range(B):
        d = cp.make_stream(ans, nf, 5200 + 17 * b + ans, payload=("random", "ring", "ring_near")[b % 3],
                           frames_per_rev=(3.3, 9.7, 25.0, 61.0, 130.0, 500.0)[b]).copy()
        if b in (1, 4):  # a few broken frames: checksum failures -> scan-reset requests mid-recording
            for f in rng.choice(nf - 2, 3, replace=False):
                d[(f + 1) * S + S // 2] ^= 0x5A
        streams.append(d)
    max_count, n_stride, scan_cap = (8192, 2048, 64) if ans != 0x83 else (8192, 64, 1024)
    # the whole recording in one pass: the expectation
    want = []
    for d in streams:
        nodes, rst, err, _ = oracle.unpack(ans, d, 125)
        scans, off = oracle.segment(nodes, rst, max_count)
        want.append([scans[off[i]: off[i + 1]] for i in range(len(off) - 1)])
    assert sum(len(w) for w in want) >= 12
    # ... and in pieces; (frames per piece differ per call, one piece shorter than a revolution)
    pieces = [37, 5, 64, 1, 90, 23, 200] if ans != 0x83 else [7, 1, 12, 3, 9, 2, 6]
    assert sum(pieces) == nf
    carry_stride = max_count
    d_carry = [torch.zeros(B, carry_stride * 8, dtype=torch.uint8, device=dev) for _ in range(2)]
    d_clen = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(2)]
    d_state = [torch.zeros(B, 4, dtype=torch.int32, device=dev) for _ in range(2)]
    got = [[] for _ in range(B)]
    at = 0
    caps = ans in (0x82, 0x84, 0x85, 0x86)
    for k, pf in enumerate(pieces):
        # a capsule piece after the first starts one frame early, state flags bit 0 set: that frame
        # only serves as the predecessor of the next one (include/rplgpu.h, rplgpu_decode_batch_dev)
        lo = at - 1 if (caps and k) else at
        buf = np.stack([d[lo * S: (at + pf) * S] for d in streams])
        at += pf
        pf = buf.shape[1] // S
        if k:
            d_state[k & 1][:, 2] = 1 if caps else 0
        d_bytes = torch.from_numpy(np.ascontiguousarray(buf)).to(dev)
        d_nf = torch.full((B,), pf, dtype=torch.int32, device=dev)
        d_batch = torch.zeros(B * scan_cap, n_stride * 8, dtype=torch.uint8, device=dev)
        d_len = torch.full((B * scan_cap,), -1, dtype=torch.int32, device=dev)
        d_ns = torch.zeros(B, dtype=torch.int32, device=dev)
        d_ne = torch.zeros(B, dtype=torch.int32, device=dev)
        d_st = torch.zeros(B, dtype=torch.int32, device=dev)
        i, o = k & 1, (k + 1) & 1
        gpu.decode_scans_carry_dev(ans, 125, d_bytes.data_ptr(), pf * S, 0, 0, d_nf.data_ptr(), pf, B,
                                   d_state[i].data_ptr(), d_state[o].data_ptr(), max_count,
                                   d_batch.data_ptr(), n_stride, scan_cap, d_len.data_ptr(),
                                   d_ns.data_ptr(), d_ne.data_ptr(), d_st.data_ptr(),
                                   d_carry[i].data_ptr() if k else 0, d_clen[i].data_ptr() if k else 0,
                                   d_carry[o].data_ptr(), d_clen[o].data_ptr(), carry_stride)
        gpu.synchronize()
        # (RPLGPU_SCAN_OUT_TRUNCATED = 8 is expected for the streams whose revolutions hold more
        # nodes than a batch slot: the slot then has the scan's first n_stride nodes)
        assert int(d_st.max()) & ~8 == 0
        batch = d_batch.cpu().numpy().view(NODE_DTYPE).reshape(B * scan_cap, n_stride)
        lens, ns = d_len.cpu().numpy(), d_ns.cpu().numpy()
        for b in range(B):
            for s_ in range(ns[b]):
                got[b].append(batch[b * scan_cap + s_, : lens[b * scan_cap + s_]].copy())
            assert np.all(lens[b * scan_cap + ns[b]: (b + 1) * scan_cap] == 0)
    for b in range(B):
        assert len(got[b]) == len(want[b]), (hex(ans), b, len(got[b]), len(want[b]))
        for j, (g, w) in enumerate(zip(got[b], want[b])):
            keep = min(len(w), n_stride)
            assert len(g) == keep and g.tobytes() == w[:keep].tobytes(), (hex(ans), b, j)
    # what is still open at the end of the recording is the oracle's unfinished scan
    clen = d_clen[len(pieces) & 1].cpu().numpy()
    for b, d in enumerate(streams):
        nodes, rst, _, _ = oracle.unpack(ans, d, 125)
        sync = np.flatnonzero(nodes["flag"] & 1)
        open_len = 0
        if len(sync) and not np.any(np.asarray(rst) > sync[-1]):
            open_len = min(len(nodes) - sync[-1], max_count)
        assert clen[b] == open_len, (hex(ans), b)


def test_decode_full_size_properties(gpu):
    """BASELINE config-3 shape for the decode stage (4096 DenseBoost streams x 801 capsules =
    131 M nodes), checked through size-independent properties computed on the device:
    every capsule but the last publishes 40 nodes; the decoded distances are the capsule words
    shifted by two (a sum of sums over the whole batch); quality is 0x2F << 2 exactly where the
    distance is non-zero; every flag byte is 1 or 2 and there is exactly one sync node per
    revolution; after scan assembly the completed scan of a stream is the node range between
    its two sync nodes."""
    torch = _torch()
    dev = torch.device("cuda:0")
    ans, nf, B = 0x85, 801, 4096
    S, npf = cp.FRAME_SIZE[ans], cp.NODES_PER_FRAME[ans]
    uniq = 16
    base = np.stack([cp.make_stream(ans, nf, 500 + s, payload="ring", frames_per_rev=nf / 2.0 + 0.3)
                     for s in range(uniq)])
    buf = torch.from_numpy(base).to(dev).repeat(B // uniq, 1).contiguous()
    d_nf = torch.full((B,), nf, dtype=torch.int32, device=dev)
    node_stride = nf * npf
    d_nodes = torch.zeros(B, node_stride * 8, dtype=torch.uint8, device=dev)
    d_nn = torch.zeros(B, dtype=torch.int32, device=dev)
    d_rst = torch.zeros(B, 8, dtype=torch.int32, device=dev)
    d_nr = torch.zeros(B, dtype=torch.int32, device=dev)
    d_st = torch.zeros(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    gpu.decode_batch_dev(ans, 125, buf.data_ptr(), nf * S, 0, 0, d_nf.data_ptr(), nf, B, 0, 0,
                         d_nodes.data_ptr(), node_stride, d_nn.data_ptr(), d_rst.data_ptr(), 8,
                         d_nr.data_ptr(), 0, d_st.data_ptr())
    gpu.synchronize()
    assert int(d_st.max()) == 0
    assert bool((d_nn == (nf - 1) * npf).all())
    assert bool((d_nr == 1).all()) and bool((d_rst[:, 0] == 0).all())  # first capsule: reset at node 0
    n = (nf - 1) * npf
    raw = d_nodes.view(B, node_stride, 8)[:, :n]
    dist = (raw[..., 2].to(torch.int64) | (raw[..., 3].to(torch.int64) << 8)
            | (raw[..., 4].to(torch.int64) << 16) | (raw[..., 5].to(torch.int64) << 24))
    cab = buf.view(B, nf, S)[:, : nf - 1, 4:].reshape(B, n, 2).to(torch.int64)
    want = (cab[..., 0] | (cab[..., 1] << 8)) << 2
    assert int(dist.sum()) == int(want.sum()) and bool((dist == want).all())
    q = raw[..., 6]
    assert bool(((q == (0x2F << 2)) == (dist != 0)).all()) and bool(((q == 0) == (dist == 0)).all())
    flag = raw[..., 7]
    assert bool(((flag == 1) | (flag == 2)).all())
    nsync = (flag == 1).sum(1)
    assert bool(((nsync >= 1) & (nsync <= 2)).all())  # two revolutions per stream, +- the start phase
    # scan assembly: one completed scan per stream with two sync nodes
    d_seg = torch.zeros_like(d_nodes)
    scan_cap = 4
    d_off = torch.zeros(B, scan_cap + 1, dtype=torch.int32, device=dev)
    d_ns = torch.zeros(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()  # the handle runs on its own stream: torch's fills must have landed
    gpu.segment_batch_dev(d_nodes.data_ptr(), node_stride, d_nn.data_ptr(), d_rst.data_ptr(), 8,
                          d_nr.data_ptr(), B, 32768, d_seg.data_ptr(), node_stride,
                          d_off.data_ptr(), scan_cap, d_ns.data_ptr(), d_st.data_ptr())
    gpu.synchronize()
    assert int(d_st.max()) == 0
    assert bool((d_ns == nsync - 1).all())
    for b in (0, 1, 17, B - 1):
        f = flag[b].cpu().numpy()
        pos = np.nonzero(f == 1)[0]
        if len(pos) == 2:
            ln = int(d_off[b, 1])
            assert ln == pos[1] - pos[0]
            assert bool((d_seg.view(B, node_stride, 8)[b, :ln] == raw[b, pos[0]: pos[1]]).all())


@pytest.fixture(scope="module")
def gpu_plain_decoder():
    """A handle whose decode calls never take the LDS-staged instance (RPLGPU_DEC_STAGE=0 is read by
    rplgpu_create): the plain kernel on the same inputs is the second reference of the test below."""
    import os

    from rplidar_ros2_driver_amd import RplGpu
    from tests.conftest import _shared_stream
    torch = _torch()
    saved = os.environ.get("RPLGPU_DEC_STAGE")
    os.environ["RPLGPU_DEC_STAGE"] = "0"
    try:
        h = RplGpu(device=0, max_samples_per_scan=32768, max_batch=64)
    finally:
        if saved is None:
            os.environ.pop("RPLGPU_DEC_STAGE", None)
        else:
            os.environ["RPLGPU_DEC_STAGE"] = saved
    h.set_stream(_shared_stream().cuda_stream)
    yield h
    torch.cuda.synchronize()
    h.close()


@pytest.mark.parametrize("ans", [0x82, 0x84, 0x85, 0x86])
@pytest.mark.parametrize("where", ["small", "limit", "beyond"])
def test_staged_decoder_matches_plain_and_oracle(gpu, gpu_plain_decoder, oracle, ans, where):
    """Back-to-back capsule streams up to rplgpu_decode_staged_frames frames are decoded out of an
    LDS copy of the stream (k_decode<..., STG>).  Same nodes, counts, reset positions, error
    counts, status and carried state as the plain kernel and as the oracle: streams of every
    length from 0 frames to the limit, payload bytes flipped (checksum failures, broken capsule
    pairs), a broken sync pattern, an odd stream stride (frames at any byte alignment), carried-in
    state, and the "frame 0 is the previous call's last frame" flag."""
    torch = _torch()
    dev = torch.device("cuda:0")
    lib = abi.load_library()
    S, npf = cp.FRAME_SIZE[ans], cp.NODES_PER_FRAME[ans]
    lim = int(lib.rplgpu_decode_staged_frames(ans))
    assert 0 < lim <= int(lib.rplgpu_decode_max_frames(ans))
    max_frames = {"small": 97, "limit": lim, "beyond": min(lim + 1, int(lib.rplgpu_decode_max_frames(ans)))}[where]
    rng = np.random.default_rng(ans * 131 + max_frames)
    B = 18
    nfs = rng.integers(3, max_frames + 1, B).astype(np.int32)
    nfs[:5] = (0, 1, 2, max_frames, max_frames)
    stride = max_frames * S + int(rng.integers(0, 3)) * 2 + 1  # odd: streams start at any alignment
    buf = np.zeros((B, stride), np.uint8)
    streams = []
    for b in range(B):
        s = cp.make_stream(ans, int(nfs[b]), 900 + 17 * b + max_frames, payload=("random", "ring", "ring_near")[b % 3],
                           frames_per_rev=float(rng.uniform(7.0, 60.0))).copy()
        if b % 4 == 1 and nfs[b] > 4:  # flipped payload bytes: checksum failures here and there
            for k in rng.integers(1, nfs[b], 1 + nfs[b] // 40):
                s[int(k) * S + int(rng.integers(4, S))] ^= 0x5A
        streams.append(s)
        buf[b, : len(s)] = s
    if nfs[7] > 6:
        buf[7, 5 * S] ^= 0xF0  # broken sync pattern: RPLGPU_STREAM_UNFRAMED, nothing published
    state_in = np.zeros((B, 4), np.int32)
    state_in[:, 0] = rng.integers(0, 2, B)
    state_in[:, 1] = rng.integers(0, 4000, B) * 4
    state_in[9:12, 2] = 1  # frame 0 only as the predecessor of frame 1
    d_bytes = torch.from_numpy(buf).to(dev)
    d_nf = torch.from_numpy(nfs).to(dev)
    d_sin = torch.from_numpy(state_in).to(dev)
    node_stride = max_frames * npf

    def run(h):
        o = {k: torch.full(shape, 0x55, dtype=dt, device=dev) for k, shape, dt in (
            ("nodes", (B, node_stride * 8), torch.uint8), ("nn", (B,), torch.int32),
            ("rst", (B, 32), torch.int32), ("nr", (B,), torch.int32), ("ne", (B,), torch.int32),
            ("st", (B,), torch.int32), ("sout", (B, 4), torch.int32))}
        o["nodes"].zero_()
        h.decode_batch_dev(ans, 125, d_bytes.data_ptr(), stride, 0, 0, d_nf.data_ptr(), max_frames, B,
                           d_sin.data_ptr(), o["sout"].data_ptr(), o["nodes"].data_ptr(), node_stride,
                           o["nn"].data_ptr(), o["rst"].data_ptr(), 32, o["nr"].data_ptr(),
                           o["ne"].data_ptr(), o["st"].data_ptr())
        h.synchronize()
        return {k: v.cpu().numpy() for k, v in o.items()}

    got, ref = run(gpu), run(gpu_plain_decoder)
    for b in range(B):
        n = int(ref["nn"][b])
        assert int(got["nn"][b]) == n and got["st"][b] == ref["st"][b], (hex(ans), where, b)
        assert got["nodes"][b, : n * 8].tobytes() == ref["nodes"][b, : n * 8].tobytes(), (hex(ans), where, b)
        assert got["nr"][b] == ref["nr"][b] and got["ne"][b] == ref["ne"][b], (hex(ans), where, b)
        k = min(int(ref["nr"][b]), 32)
        assert list(got["rst"][b, :k]) == list(ref["rst"][b, :k]), (hex(ans), where, b)
        assert list(got["sout"][b]) == list(ref["sout"][b]), (hex(ans), where, b)
        if b == 7 and nfs[7] > 6:
            assert got["st"][b] & abi_status("UNFRAMED") and n == 0
            continue
        if state_in[b, 2]:
            continue  # (the cut-recording flag: plain kernel is the reference, oracle-checked elsewhere)
        want, w_rst, w_err, w_st = oracle.unpack(ans, buf[b, : int(nfs[b]) * S], 125,
                                                 state=(int(state_in[b, 0]), int(state_in[b, 1])))
        assert n == len(want) and got["st"][b] == 0, (hex(ans), where, b)
        nodes = got["nodes"][b].view(NODE_DTYPE)[:n]
        assert nodes.tobytes() == want.tobytes(), (hex(ans), where, b)
        assert list(got["rst"][b, : got["nr"][b]]) == list(w_rst) and int(got["ne"][b]) == w_err
        assert tuple(int(v) for v in got["sout"][b, :2]) == w_st


@pytest.mark.parametrize("ans", [0x82, 0x84, 0x85, 0x86])
def test_staged_decoder_with_frame_offsets(gpu, gpu_plain_decoder, oracle, ans):
    """The same with frame offsets and gap flags (rplgpu_frame_stream's output): the staged kernel
    copies [first frame, end of last frame) and takes a stream only when every frame lies inside
    that copy at most 16383 rejected bytes behind its back-to-back place; the others (a huge hole,
    a span longer than the LDS the call reserved, offsets that go backwards) are listed for the
    plain kernel.  Either way: the plain kernel's and the oracle's nodes, counts and state."""
    torch = _torch()
    dev = torch.device("cuda:0")
    lib = abi.load_library()
    S, npf = cp.FRAME_SIZE[ans], cp.NODES_PER_FRAME[ans]
    lim = int(lib.rplgpu_decode_staged_frames(ans))
    max_frames = min(lim, 300)
    rng = np.random.default_rng(ans * 977 + 5)
    B = 14
    nfs = rng.integers(3, max_frames + 1, B).astype(np.int32)
    nfs[:4] = (0, 1, 2, max_frames)
    offs = np.zeros((B, max_frames), np.uint32)
    gaps = np.zeros((B, max_frames), np.uint8)
    blobs = []
    for b in range(B):
        nf = int(nfs[b])
        fr = cp.make_frames(ans, nf, 4000 + 13 * b, payload=("random", "ring", "ring_noisy")[b % 3],
                            frames_per_rev=float(rng.uniform(7.0, 40.0)))
        if b % 4 == 1 and nf > 4:
            for k in rng.integers(1, nf, 1 + nf // 30):
                fr[int(k), int(rng.integers(4, S))] ^= 0x3C
        # rejected bytes in front of some frames (flagged as gaps, as the framer would)
        junk = np.where(rng.random(nf) < 0.15, rng.integers(1, 40, nf), 0).astype(np.int64)
        if nf:
            junk[0] = int(rng.integers(0, 7))
        if b == 5 and nf > 10:
            junk[7] = 20000        # a hole the table entry cannot hold: plain kernel
        if b == 6 and nf > 10:
            junk[3:] += 400        # span beyond the reserved LDS for a long stream: plain kernel (or fits: fine)
        pos = np.cumsum(junk + S) - S
        if b == 7 and nf > 10:     # offsets that go backwards: plain kernel
            pos[[4, 5]] = pos[[5, 4]]
        blob = np.zeros(int(pos.max() + S) if nf else 1, np.uint8)
        blob[:] = rng.integers(0, 256, len(blob), dtype=np.uint8)
        for k in range(nf):
            blob[pos[k]: pos[k] + S] = fr[k]
        offs[b, :nf] = pos
        gaps[b, :nf] = (junk > 0) & (np.arange(nf) > 0)
        blobs.append(blob)
    stride = max(len(x) for x in blobs) + 3
    buf = np.zeros((B, stride), np.uint8)
    for b, x in enumerate(blobs):
        buf[b, : len(x)] = x
    state_in = np.zeros((B, 4), np.int32)
    state_in[:, 0] = rng.integers(0, 2, B)
    state_in[:, 1] = rng.integers(0, 4000, B) * 4
    d_bytes, d_nf = torch.from_numpy(buf).to(dev), torch.from_numpy(nfs).to(dev)
    d_off, d_gap = torch.from_numpy(offs).to(dev), torch.from_numpy(gaps).to(dev)
    d_sin = torch.from_numpy(state_in).to(dev)
    node_stride = max_frames * npf

    def run(h):
        o = {k: torch.full(shape, 0x55, dtype=dt, device=dev) for k, shape, dt in (
            ("nodes", (B, node_stride * 8), torch.uint8), ("nn", (B,), torch.int32),
            ("rst", (B, 32), torch.int32), ("nr", (B,), torch.int32), ("ne", (B,), torch.int32),
            ("st", (B,), torch.int32), ("sout", (B, 4), torch.int32))}
        o["nodes"].zero_()
        h.decode_batch_dev(ans, 125, d_bytes.data_ptr(), stride, d_off.data_ptr(), d_gap.data_ptr(),
                           d_nf.data_ptr(), max_frames, B, d_sin.data_ptr(), o["sout"].data_ptr(),
                           o["nodes"].data_ptr(), node_stride, o["nn"].data_ptr(), o["rst"].data_ptr(), 32,
                           o["nr"].data_ptr(), o["ne"].data_ptr(), o["st"].data_ptr())
        h.synchronize()
        return {k: v.cpu().numpy() for k, v in o.items()}

    got, ref = run(gpu), run(gpu_plain_decoder)
    for b in range(B):
        n = int(ref["nn"][b])
        assert int(got["nn"][b]) == n and got["st"][b] == ref["st"][b], (hex(ans), b)
        assert got["nodes"][b, : n * 8].tobytes() == ref["nodes"][b, : n * 8].tobytes(), (hex(ans), b)
        assert got["nr"][b] == ref["nr"][b] and got["ne"][b] == ref["ne"][b], (hex(ans), b)
        k = min(int(ref["nr"][b]), 32)
        assert list(got["rst"][b, :k]) == list(ref["rst"][b, :k]), (hex(ans), b)
        assert list(got["sout"][b]) == list(ref["sout"][b]), (hex(ans), b)
        if b == 7:
            continue  # (offsets out of order: the oracle walks them in the order given too, but keep it simple)
        nf = int(nfs[b])
        want, w_rst, w_err, w_st = oracle.unpack_frames(ans, buf[b], offs[b, :nf], gaps[b, :nf], 125,
                                                        state=(int(state_in[b, 0]), int(state_in[b, 1])))
        assert n == len(want), (hex(ans), b)
        assert got["nodes"][b].view(NODE_DTYPE)[:n].tobytes() == want.tobytes(), (hex(ans), b)
        assert list(got["rst"][b, : got["nr"][b]]) == list(w_rst) and int(got["ne"][b]) == w_err
        assert tuple(int(v) for v in got["sout"][b, :2]) == w_st
