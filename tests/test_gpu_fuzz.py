"""Seeded differential fuzzing of the HIP path against the CPU oracle: random scans (sizes,
angle orders, distance distributions chosen to sit on the awkward spots — cell faces, clip
bounds, the u32 range) x random parameters.  Same bars as tests/test_gpu_parity.py: integer
work bit-exact, floats bit-exact where the design makes them so, voxel centroids within 1e-6 m.
Angles are unique within a scan wherever the comparison is exact (upstream's std::sort is
unstable, so ties have no single right answer; tests/test_gpu_parity.py covers them)."""
import os

import numpy as np
import pytest

from rplidar_ros2_driver_amd import NODE_DTYPE, Params
from tests import oracle_lib

pytestmark = pytest.mark.gpu

XYZ_TOL = 1e-6
LEAVES = [0.05, 0.05, 0.01, 0.1, 0.25, 1.0 / 3.0, 0.02, 1.0, 0.0625]


def _random_scan(rng):
    n = int(rng.choice([1, 2, 3, 7, 64, 65, 360, 1000, 1023, 1025, 2500, int(rng.integers(1, 4000))]))
    nodes = np.zeros(n, NODE_DTYPE)
    q = rng.choice(65536, size=n, replace=False)  # unique angle words
    order = rng.integers(0, 3)
    if order == 0:
        q = np.sort(q)
    elif order == 1:  # rotated ascending (a scan that starts mid-revolution)
        q = np.roll(np.sort(q), int(rng.integers(0, n)))
    nodes["angle_z_q14"] = q
    kind = rng.integers(0, 6)
    if kind == 0:    # anything a u32 can hold
        d = rng.integers(0, 2**32, n, dtype=np.uint64)
    elif kind == 1:  # a ring with range noise
        r = rng.uniform(0.2, 39.0) + rng.normal(0, rng.choice([0.0, 0.002, 0.02, 0.2]), n)
        d = np.clip(r * 4000.0, 0, 2**32 - 1).astype(np.uint64)
    elif kind == 2:  # multiples of the leaf along the axes: points on cell faces
        leaf_q2 = int(rng.choice([200, 40, 400, 1000]))
        d = (rng.integers(1, 600, n) * leaf_q2 + rng.integers(-1, 2, n)).astype(np.uint64)
        nodes["angle_z_q14"] = rng.permutation(np.r_[np.arange(0, 65536, 16384),
                                                      rng.choice(65536, size=max(n - 4, 0))])[:n] \
            if n >= 4 else nodes["angle_z_q14"]
    elif kind == 3:  # around the clip bounds 0.15 m / range_max
        d = rng.choice([599, 600, 601, 47999, 48000, 48001, 159999, 160000, 160001, 1, 2], n)
    elif kind == 4:  # close range: many samples per cell
        d = rng.integers(600, 4000, n)
    else:            # uniform over the sensor's range
        d = rng.integers(600, 160001, n)
    d = np.asarray(d, dtype=np.uint64)
    drop = rng.random(n) < rng.choice([0.0, 0.1, 0.5, 0.95])
    d[drop] = 0
    nodes["dist_mm_q2"] = d.astype(np.uint32)
    nodes["quality"] = rng.integers(0, 256, n)
    nodes["flag"] = rng.integers(0, 4, n)
    return nodes


def _unique_valid_angles(nodes):
    v = nodes[nodes["dist_mm_q2"] != 0]
    return len(np.unique(v["angle_z_q14"])) == len(v)


# RPL_FUZZ_SEEDS=400 for a long run (25 scans x parameter draws per seed)
@pytest.mark.parametrize("seed", range(int(os.environ.get("RPL_FUZZ_SEEDS", "48"))))
def test_fuzz_against_oracle(gpu, oracle, seed):
    rng = np.random.default_rng(9000 + seed)
    for it in range(25):
        nodes = _random_scan(rng)
        unique = _unique_valid_angles(nodes)
        is_new, inverted, mode = (int(x) for x in rng.integers(0, 2, 3))
        clip = int(rng.integers(0, 2))
        rmax = float(rng.choice([12.0, 40.0, 8.0, 0.3, 1.0e9]))
        rmin = float(rng.choice([0.15, 0.0, 0.5, 12.0]))
        qmin = int(rng.choice([0, 1, 40, 255]))
        ctx = f"seed {seed} it {it} n {len(nodes)}"

        # S1 ascend
        asc = nodes.copy()
        res = gpu.ascend(asc)
        want, wres = oracle.ascend(nodes)
        assert res == wres, ctx
        if res == 0:
            assert oracle_lib.canon_equal_angle_runs(asc).tobytes() == \
                oracle_lib.canon_equal_angle_runs(want).tobytes(), ctx
        else:
            assert asc.tobytes() == nodes.tobytes(), ctx

        # S3 publish_scan (+ E1 clip)
        p = Params.defaults(is_new_protocol=is_new, inverted=inverted, scan_processing=mode,
                            clip_enable=clip, q_min=qmin, range_min=rmin, range_max=rmax)
        wr, wi, wm = oracle.publish_scan(nodes, oracle_lib.copy_params(p), 0.0731)
        gr, gi, gm = gpu.scan_to_laserscan(nodes, p, 0.0731)
        assert bytes(gm) == bytes(wm), ctx
        if wm.published:
            if mode or unique:
                assert gr.tobytes() == wr.tobytes(), ctx
            if unique:
                assert gi.tobytes() == wi.tobytes(), ctx

        # ext: cloud, optionally ROR, optionally voxel
        leaf = float(LEAVES[int(rng.integers(0, len(LEAVES)))])
        voxel = int(rng.integers(0, 2))
        ror = int(rng.integers(0, 3) == 0)
        pc = Params.defaults(is_new_protocol=is_new, inverted=inverted, clip_enable=1, q_min=qmin,
                             range_min=rmin, range_max=min(rmax, 40.0), voxel_enable=voxel,
                             voxel_leaf=leaf, ror_enable=ror,
                             ror_radius=float(rng.choice([0.1, 0.05, 0.5])),
                             ror_min_neighbors=int(rng.integers(1, 4)))
        got, status = gpu.scan_to_cloud(nodes, pc, allow_overflow=True)
        assert status == 0, ctx  # ranges <= 40 m and leaf >= 1 cm: |cell| < 32767
        if voxel or ror:
            wantc, wcells, _ = oracle.cloud_pipeline(nodes, oracle_lib.copy_params(pc))
        else:
            wantc = oracle.scan_to_cloud(nodes, oracle_lib.copy_params(pc))
        assert len(got) == len(wantc), ctx
        if len(wantc):
            if voxel:
                assert np.max(np.abs(got[:, :2].astype(np.float64) - wantc[:, :2])) <= XYZ_TOL, ctx
                assert np.all(got[:, 2] == 0.0), ctx
            else:
                assert got[:, :3].tobytes() == wantc[:, :3].tobytes(), ctx
            assert got[:, 3].tobytes() == wantc[:, 3].tobytes(), ctx


@pytest.mark.parametrize("seed", range(int(os.environ.get("RPL_FUZZ_SEEDS", "48"))))
def test_fuzz_decode_against_oracle(gpu, oracle, seed):
    """Recorded answer streams of all six types: random lengths, heavy fault rates, pure byte
    soup, and recordings cut at arbitrary frame boundaries with the state carried over —
    nodes, reset positions, error counts and carried state bit for bit."""
    from rplidar_ros2_driver_amd import capsules as cp
    rng = np.random.default_rng(77000 + seed)
    for ans in (0x81, 0x82, 0x83, 0x84, 0x85, 0x86):
        for it in range(3):
            nf = int(rng.choice([1, 2, 3, 17, 64, 65, int(rng.integers(1, 400))]))
            if ans == 0x81:
                nf *= 8
            dur = int(rng.choice([125, 32, 20, 2, 1000000, 476]))
            state = (int(rng.integers(0, 2)), int(rng.integers(0, 2)) * int(rng.integers(0, 40000)))
            mode = int(rng.integers(0, 4))
            frames = cp.make_frames(ans, nf, int(rng.integers(0, 1 << 30)),
                                    payload=str(rng.choice(["random", "ring", "ring_near", "ring_noisy"])),
                                    frames_per_rev=float(rng.choice([1.5, 3.1, 12.3, 40.0, 300.0])),
                                    first_sync=bool(rng.integers(0, 2)))
            if mode == 0:
                data = frames.reshape(-1)
            elif mode == 1:
                data = cp.corrupt_stream(ans, frames, int(rng.integers(0, 1 << 30)))
            elif mode == 2:  # a bad link: a fault in every third frame or so
                data = cp.corrupt_stream(ans, frames, int(rng.integers(0, 1 << 30)),
                                         p_checksum=0.3, p_sync=0.2, p_garbage=0.2,
                                         p_revstart=0.2, p_jump=0.2)
            else:            # no framing at all
                data = rng.integers(0, 256, int(rng.integers(0, 4000)), dtype=np.uint8)
            ctx = f"seed {seed} ans {ans:#x} it {it} nf {nf} mode {mode}"
            nodes, rst, err, st = gpu.decode_stream(ans, data, dur, state)
            w_nodes, w_rst, w_err, w_st = oracle.unpack(ans, data, dur, state=state)
            assert len(nodes) == len(w_nodes), ctx
            assert nodes.tobytes() == w_nodes.tobytes(), ctx
            assert list(rst) == list(w_rst) and err == w_err and st == w_st, ctx
            _fuzz_decode_unframed(gpu, oracle, ans, frames, dur, state, rng, ctx)
            _fuzz_decode_scans(gpu, oracle, ans, data, dur, state, w_nodes, w_rst, w_err, w_st, rng, ctx)
            _fuzz_decode_scans_carry(gpu, oracle, ans, data, dur, state, w_nodes, w_rst, w_err, w_st, rng, ctx)
            # the same recording in two pieces, state handed over (cut on a frame boundary of
            # the clean stream; for faulty streams the cut lands anywhere, which is the point)
            S = cp.FRAME_SIZE[ans]
            cut = int(rng.integers(0, max(len(data) // S, 1) + 1)) * S
            if mode == 0 and 0 < cut < len(data):
                n1, r1, e1, s1 = gpu.decode_stream(ans, data[:cut], dur, state)
                w1 = oracle.unpack(ans, data[:cut], dur, state=state)
                assert n1.tobytes() == w1[0].tobytes() and list(r1) == list(w1[1]), ctx
                assert e1 == w1[2] and s1 == w1[3], ctx


def _fuzz_decode_unframed(gpu, oracle, ans, frames, dur, state, rng, ctx):
    """The frames back to back, no offsets (rplgpu_decode_batch_dev with d_frame_off = NULL: the
    LDS-staged decoder for the capsule types), payload and header bytes flipped but the sync
    nibbles left alone so that the byte machine of the oracle finds the same frames."""
    import torch
    from rplidar_ros2_driver_amd import capsules as cp
    if ans not in (0x82, 0x84, 0x85, 0x86):
        return
    S, npf = cp.FRAME_SIZE[ans], cp.NODES_PER_FRAME[ans]
    f = frames.copy().reshape(-1, S)
    nf = len(f)
    for _ in range(int(rng.integers(0, 1 + nf // 8))):
        k, at = int(rng.integers(0, nf)), int(rng.integers(0, S))
        f[k, at] ^= np.uint8(int(rng.integers(1, 256)) & (0x0F if at < 2 else 0xFF))
    data = f.reshape(-1)
    want, w_rst, w_err, w_st = oracle.unpack(ans, data, dur, state=state)
    dev = torch.device("cuda:0")
    pad = int(rng.integers(0, 4))  # the stream starts at any byte alignment
    d_bytes = torch.from_numpy(np.concatenate([np.zeros(pad, np.uint8), data])).to(dev)
    d_nf = torch.tensor([nf], dtype=torch.int32, device=dev)
    d_sin = torch.tensor([[state[0], state[1], 0, 0]], dtype=torch.int32, device=dev)
    node_stride = nf * npf
    d_nodes = torch.zeros(node_stride * 8, dtype=torch.uint8, device=dev)
    d_i = torch.zeros(5, dtype=torch.int32, device=dev)  # n_nodes, n_reset, n_errors, status
    d_rst = torch.zeros(nf + 2, dtype=torch.int32, device=dev)
    d_sout = torch.zeros(4, dtype=torch.int32, device=dev)
    gpu.decode_batch_dev(ans, dur, d_bytes.data_ptr() + pad, len(data), 0, 0, d_nf.data_ptr(), nf, 1,
                         d_sin.data_ptr(), d_sout.data_ptr(), d_nodes.data_ptr(), node_stride,
                         d_i.data_ptr(), d_rst.data_ptr(), nf + 2, d_i.data_ptr() + 4,
                         d_i.data_ptr() + 8, d_i.data_ptr() + 12)
    gpu.synchronize()
    n, nr, ne, st = (int(v) for v in d_i.cpu().numpy()[:4])
    assert st == 0 and n == len(want), ctx
    assert d_nodes.cpu().numpy()[: n * 8].tobytes() == want.tobytes(), ctx
    assert list(d_rst.cpu().numpy()[:nr]) == list(w_rst) and ne == w_err, ctx
    assert tuple(int(v) for v in d_sout.cpu().numpy()[:2]) == w_st, ctx


def _fuzz_decode_scans(gpu, oracle, ans, data, dur, state, w_nodes, w_rst, w_err, w_st, rng, ctx):
    """The same stream through rplgpu_decode_scans_dev (host framing; the fused decoder for express /
    ultra / dense, the general path otherwise): completed scans in batch slots == oracle.segment of
    the oracle's nodes, error count and carried state unchanged."""
    import torch
    from rplidar_ros2_driver_amd import NODE_DTYPE, abi
    from rplidar_ros2_driver_amd import capsules as cp
    dev = torch.device("cuda:0")
    off, gap = abi.frame_stream(ans, data)
    max_frames = abi.load_library().rplgpu_decode_max_frames(ans)
    if len(off) == 0 or len(off) > max_frames:
        return
    nf = len(off)
    max_count = int(rng.choice([8192, 37, 5]))
    n_stride = int(rng.choice([4096, 64, 9]))
    scan_cap = int(rng.choice([512, 4, 1]))
    d_bytes = torch.from_numpy(np.ascontiguousarray(data)).to(dev)
    d_off = torch.from_numpy(np.asarray(off, np.uint32).view(np.int32)).to(dev)
    d_gap = torch.from_numpy(np.asarray(gap, np.uint8)).to(dev)
    d_nf = torch.tensor([nf], dtype=torch.int32, device=dev)
    d_sin = torch.tensor([state[0], state[1], 0, 0], dtype=torch.int32, device=dev)
    d_sout = torch.zeros(4, dtype=torch.int32, device=dev)
    d_batch = torch.zeros(scan_cap, n_stride * 8, dtype=torch.uint8, device=dev)
    d_len = torch.full((scan_cap,), -1, dtype=torch.int32, device=dev)
    d_ns = torch.zeros(1, dtype=torch.int32, device=dev)
    d_ne = torch.zeros(1, dtype=torch.int32, device=dev)
    d_st = torch.zeros(1, dtype=torch.int32, device=dev)
    gpu.decode_scans_dev(ans, dur, d_bytes.data_ptr(), len(data), d_off.data_ptr(), d_gap.data_ptr(),
                         d_nf.data_ptr(), nf, 1, d_sin.data_ptr(), d_sout.data_ptr(), max_count,
                         d_batch.data_ptr(), n_stride, scan_cap, d_len.data_ptr(), d_ns.data_ptr(),
                         d_ne.data_ptr(), d_st.data_ptr())
    gpu.synchronize()
    scans, w_off = oracle.segment(w_nodes, w_rst, max_count)
    n_want = len(w_off) - 1
    lens = d_len.cpu().numpy()
    batch = d_batch.cpu().numpy().view(NODE_DTYPE).reshape(scan_cap, n_stride)
    assert int(d_ne.item()) == w_err and int(d_ns.item()) == min(n_want, scan_cap), ctx
    assert tuple(int(v) for v in d_sout.cpu().numpy()[:2]) == tuple(w_st), ctx
    for s_ in range(scan_cap):
        if s_ >= n_want:
            assert lens[s_] == 0, ctx
            continue
        scan = scans[w_off[s_]: w_off[s_ + 1]]
        keep = min(len(scan), n_stride)
        assert lens[s_] == keep and batch[s_, :keep].tobytes() == scan[:keep].tobytes(), (ctx, s_)


def _fuzz_decode_scans_carry(gpu, oracle, ans, data, dur, state, w_nodes, w_rst, w_err, w_st, rng, ctx):
    """The same stream through rplgpu_decode_scans_carry_dev in random pieces (host framing, pieces cut
    at arbitrary frames, a capsule piece after the first one frame early with flags bit 0, decoder
    state and the open scan handed from call to call, random ScanDataHolder capacity): the scans of
    all pieces together == oracle.segment of the oracle's nodes for the WHOLE stream."""
    import torch
    from rplidar_ros2_driver_amd import NODE_DTYPE, abi
    dev = torch.device("cuda:0")
    off, gap = abi.frame_stream(ans, data)
    max_frames = abi.load_library().rplgpu_decode_max_frames(ans)
    nf = len(off)
    if nf < 2 or nf > max_frames:
        return
    max_count = int(rng.choice([8192, 37, 5]))
    scans, w_off = oracle.segment(w_nodes, w_rst, max_count)
    n_want = len(w_off) - 1
    scan_cap = 512
    if n_want > scan_cap or len(w_nodes) > 200000:
        return
    n_stride = int(min(max(np.diff(w_off).max() if n_want else 1, 1), 8192))
    caps = ans in (0x82, 0x84, 0x85, 0x86)
    cuts = sorted(set(int(c) for c in rng.integers(1, nf, int(rng.integers(1, 5)))))
    bounds = [0] + cuts + [nf]
    d_bytes = torch.from_numpy(np.ascontiguousarray(data)).to(dev)
    off = np.asarray(off, np.uint32)
    gap = np.asarray(gap, np.uint8)
    d_carry = [torch.zeros(max(max_count, 1) * 8, dtype=torch.uint8, device=dev) for _ in range(2)]
    d_clen = [torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(2)]
    d_state = [torch.tensor([state[0], state[1], 0, 0], dtype=torch.int32, device=dev),
               torch.zeros(4, dtype=torch.int32, device=dev)]
    got, errs = [], 0
    for k in range(len(bounds) - 1):
        first, end = bounds[k], bounds[k + 1]
        lo = first - 1 if (caps and k) else first
        cnt = end - lo
        d_off = torch.from_numpy(np.ascontiguousarray(off[lo:end]).view(np.int32)).to(dev)
        d_gap = torch.from_numpy(np.ascontiguousarray(gap[lo:end])).to(dev)
        d_nf = torch.tensor([cnt], dtype=torch.int32, device=dev)
        i, o = k & 1, (k + 1) & 1
        d_state[i][2] = 1 if (caps and k) else 0
        d_batch = torch.zeros(scan_cap, n_stride * 8, dtype=torch.uint8, device=dev)
        d_len = torch.full((scan_cap,), -1, dtype=torch.int32, device=dev)
        d_ns = torch.zeros(1, dtype=torch.int32, device=dev)
        d_ne = torch.zeros(1, dtype=torch.int32, device=dev)
        d_st = torch.zeros(1, dtype=torch.int32, device=dev)
        gpu.decode_scans_carry_dev(ans, dur, d_bytes.data_ptr(), len(data), d_off.data_ptr(), d_gap.data_ptr(),
                                   d_nf.data_ptr(), cnt, 1, d_state[i].data_ptr(), d_state[o].data_ptr(),
                                   max_count, d_batch.data_ptr(), n_stride, scan_cap, d_len.data_ptr(),
                                   d_ns.data_ptr(), d_ne.data_ptr(), d_st.data_ptr(),
                                   d_carry[i].data_ptr() if k else 0, d_clen[i].data_ptr() if k else 0,
                                   d_carry[o].data_ptr(), d_clen[o].data_ptr(), max(max_count, 1))
        gpu.synchronize()
        assert int(d_st.item()) & ~8 == 0, (ctx, k, int(d_st.item()))
        errs += int(d_ne.item())
        lens = d_len.cpu().numpy()
        batch = d_batch.cpu().numpy().view(NODE_DTYPE).reshape(scan_cap, n_stride)
        for s_ in range(int(d_ns.item())):
            got.append(batch[s_, : lens[s_]].copy())
    last = len(bounds) - 1
    assert errs == w_err, (ctx, "errors")
    assert tuple(int(v) for v in d_state[last & 1].cpu().numpy()[:2]) == tuple(w_st), (ctx, "state")
    assert len(got) == n_want, (ctx, len(got), n_want, bounds)
    for j, g in enumerate(got):
        want = scans[w_off[j]: w_off[j + 1]]
        keep = min(len(want), n_stride)
        assert len(g) == keep and g.tobytes() == want[:keep].tobytes(), (ctx, j, bounds)

