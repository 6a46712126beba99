"""GPU parity at the BENCH scale and on the other data regimes.

``bench.py`` measures BASELINE config 3 — seed 2026, 4096 scans x 32 000 samples through
``rplgpu_cloud_arena_dev`` — and reports noisy / uniform variants of the same shape.  These
tests run exactly those batches through the same entry point and compare with the CPU oracle:
  * EVERY scan in full (number of cells, the (iy, ix) key of every point bit-exact and in the
    oracle's order, centroids <= 1e-6 m, mean intensity bit-exact; the oracle runs the whole
    batch on all host cores, oracle/oracle.cpp orc_batch_cloud_check),
  * the same call without the optional cell-key output (the production form) byte for byte,
  * ``status_bits == 0`` and a gap-free arena.
E1-E5 are not in the reference: the oracle is the spec of SURVEY.md §8(a-ext) ("parity
unpinned")."""
import ctypes as C
import os

import numpy as np
import pytest

from rplidar_ros2_driver_amd import Params, synth
from tests import oracle_lib

pytestmark = pytest.mark.gpu

XYZ_TOL = 1e-6


def _run_arena(gpu, batch, p, cap_per_scan, with_keys=True):
    import torch
    dev = torch.device("cuda:0")
    B, n = batch.shape
    d_nodes = torch.from_numpy(batch.view(np.uint8).reshape(B, n * 8)).to(dev)
    d_len = torch.full((B,), n, dtype=torch.int32, device=dev)
    cap = B * cap_per_scan
    d_arena = torch.empty(cap, 4, dtype=torch.float32, device=dev)
    d_cur = torch.zeros(1, dtype=torch.int64, device=dev)
    d_start = torch.zeros(B, dtype=torch.int64, device=dev)
    d_np = torch.zeros(B, dtype=torch.int32, device=dev)
    d_st = torch.zeros(B, dtype=torch.int32, device=dev)
    # the optional cell-key output (include/rplgpu.h): one word per output point, same index.
    # (It changes how a multi-band scan reaches the arena — count, reserve, write instead of the
    # temporary cell area + one copy — so every batch is run both ways and the bytes compared.)
    d_keys = torch.zeros(cap if with_keys else 1, dtype=torch.int32, device=dev)
    if with_keys:
        gpu.set_cell_key_output(d_keys.data_ptr())
    try:
        gpu.cloud_arena_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, p, d_arena.data_ptr(), cap,
                            d_cur.data_ptr(), d_start.data_ptr(), d_np.data_ptr(), d_st.data_ptr())
        gpu.synchronize()
    finally:
        gpu.set_cell_key_output(0)
    total = int(d_cur.item())
    return (d_arena[:total].cpu().numpy(), d_start.cpu().numpy(),
            d_np.cpu().numpy().astype(np.int64), d_st.cpu().numpy(), total,
            d_keys[:total].cpu().numpy().view(np.uint32) if with_keys else None)


def _check_batch(gpu, oracle, batch, p, cap_per_scan):
    """EVERY scan of the batch against the oracle (all host cores): number of cells, the (iy, ix)
    key of every output point bit-exact and in the oracle's order, mean intensity bit-exact, z = 0,
    centroids within 1e-6 m."""
    B, n = batch.shape
    arena, start, npts, st, total, keys = _run_arena(gpu, batch, p, cap_per_scan)
    assert int(st.max()) == 0, "status_bits"
    assert total == int(npts.sum())
    # the reservations tile [0, total) without gaps or overlaps
    nz = np.nonzero(npts)[0]
    order = nz[np.argsort(start[nz], kind="stable")]
    assert np.array_equal(np.cumsum(npts[order]) - npts[order], start[order])
    op = oracle_lib.copy_params(p)
    bad, res = oracle.batch_cloud_check(batch, op, arena, start, npts, keys, os.cpu_count() or 1)
    first = np.nonzero((res[:, 1] != 0) | (res[:, 2] != 0))[0]
    assert bad == 0, (bad, first[:8], res[first[:8]])
    assert np.array_equal(res[:, 0].astype(np.int64), npts)
    worst = float(res[:, 3].copy().view(np.float32).max())
    assert worst <= XYZ_TOL, worst
    # keys strictly ascending inside every scan: (iy, ix) order, no cell twice
    d = np.diff(keys.astype(np.int64))
    inner = np.ones(total - 1, bool) if total > 1 else np.zeros(0, bool)
    ends = (start + npts)[npts > 0]
    inner[ends[ends < total] - 1] = False  # (the step from one scan's last cell to the next scan's first)
    assert np.all(d[inner] > 0)
    # the production form of the same call (no cell-key output): the same cloud, byte for byte,
    # scan by scan (the scans sit in the arena in completion order, which differs between runs)
    arena2, start2, npts2, st2, total2, _ = _run_arena(gpu, batch, p, cap_per_scan, with_keys=False)
    assert total2 == total and int(st2.max()) == 0 and np.array_equal(npts2, npts)
    a1 = np.ascontiguousarray(arena).view(np.uint8).reshape(-1, 16)
    a2 = np.ascontiguousarray(arena2).view(np.uint8).reshape(-1, 16)
    idx1 = np.concatenate([np.arange(start[b], start[b] + npts[b]) for b in range(B)]) if total else np.zeros(0, np.int64)
    idx2 = np.concatenate([np.arange(start2[b], start2[b] + npts2[b]) for b in range(B)]) if total else np.zeros(0, np.int64)
    assert a1[idx1].tobytes() == a2[idx2].tobytes()
    return total, worst


def test_bench_batch_config3_voxel_matches_oracle(gpu_mode, oracle):
    gpu = gpu_mode
    """The bench batch itself: seed 2026, 4096 x 32 000, the bench parameters."""
    B, n = 4096, 32000
    batch = synth.make_batch(2026, B, n)
    p = Params.defaults(clip_enable=1, q_min=0, range_min=0.15, range_max=40.0, voxel_enable=1,
                        voxel_leaf=0.05)
    total, worst = _check_batch(gpu, oracle, batch, p, cap_per_scan=8192)
    assert total > 9_000_000  # ~2.7 k cells per scan
    # the same batch with the quality filter of BASELINE config 3 switched on
    pq = Params.defaults(clip_enable=1, q_min=48, range_min=0.15, range_max=40.0, voxel_enable=1,
                         voxel_leaf=0.05)
    total_q, _ = _check_batch(gpu, oracle, batch[:512], pq, cap_per_scan=8192)
    assert 0 < total_q


@pytest.mark.parametrize("regime", ["ring_noise_1cm", "uniform"])
def test_other_regimes_at_full_scan_size(gpu_mode, oracle, regime):
    gpu = gpu_mode
    """256 scans x 32 000 samples of the two other generators of SURVEY.md §8(d): a ring with
    1 cm range noise (what a real lidar delivers: neighbouring samples alternate between cells)
    and uniformly random ranges (nearly every sample its own cell)."""
    B, n = 256, 32000
    if regime == "uniform":
        batch = synth.make_batch(2026, B, n, kind="uniform")
    else:
        batch = synth.make_batch(2026, B, n, noise_m=0.01)
    p = Params.defaults(clip_enable=1, q_min=0, range_min=0.15, range_max=40.0, voxel_enable=1,
                        voxel_leaf=0.05)
    _check_batch(gpu, oracle, batch, p, cap_per_scan=n)


def test_c5_shape_ror_voxel_batch_matches_oracle(gpu_mode, oracle):
    gpu = gpu_mode
    """Config 5 at batch scale: 8 sensors x 8 frames of 32 000 noisy samples, E5 + E4."""
    B, n = 64, 32000
    batch = synth.make_batch(2031, B, n, noise_m=0.01)
    p = Params.defaults(clip_enable=1, range_min=0.15, range_max=40.0, voxel_enable=1,
                        voxel_leaf=0.05, ror_enable=1, ror_radius=0.10, ror_min_neighbors=2)
    _check_batch(gpu, oracle, batch, p, cap_per_scan=n)


def test_every_sample_its_own_run_at_the_largest_scan(gpu_mode, oracle):
    """The worst case of the record budget: scans of the maximum length whose samples are
    all valid and all in different cells, so that every sample ends a run and every block adds its
    marker entries (two per block when the block is aggregated in two classes) — a single scan per
    work item and a fused group of four sharing one grid."""
    import torch
    gpu = gpu_mode
    dev = torch.device("cuda:0")
    B, n = 8, 32768
    batch = synth.make_batch(77, B, n, kind="uniform", invalid_p=0.0)
    p = Params.defaults(clip_enable=1, q_min=0, range_min=0.15, range_max=40.0, voxel_enable=1,
                        voxel_leaf=0.05)
    _check_batch(gpu, oracle, batch, p, cap_per_scan=n)
    # the same scans as two groups of four, identity poses, no motion: one grid per group
    group = 4
    ng = B // group
    d_nodes = torch.from_numpy(batch.view(np.uint8).reshape(B, n * 8)).to(dev)
    d_len = torch.full((B,), n, dtype=torch.int32, device=dev)
    cap = B * n
    d_arena = torch.zeros(cap, 4, dtype=torch.float32, device=dev)
    d_cur = torch.zeros(1, dtype=torch.int64, device=dev)
    d_start = torch.zeros(ng, dtype=torch.int64, device=dev)
    d_np = torch.zeros(ng, dtype=torch.int32, device=dev)
    d_st = torch.zeros(ng, dtype=torch.int32, device=dev)
    gpu.cloud_fused_voxel_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, group, p, 0, 0,
                              d_arena.data_ptr(), cap, d_cur.data_ptr(), d_start.data_ptr(),
                              d_np.data_ptr(), d_st.data_ptr())
    gpu.synchronize()
    assert int(d_st.max()) == 0
    arena, start, npts = d_arena.cpu().numpy(), d_start.cpu().numpy(), d_np.cpu().numpy()
    op = oracle_lib.copy_params(p)
    op.voxel_enable = 0
    for g in range(ng):
        pts = np.concatenate([oracle.scan_to_cloud(batch[b], op) for b in range(g * group, (g + 1) * group)])
        want, _, _ = oracle.voxel_grid(pts, 0.05)
        got = arena[start[g]: start[g] + npts[g]]
        assert len(got) == len(want), g
        assert np.max(np.abs(got[:, :2].astype(np.float64) - want[:, :2])) <= XYZ_TOL
        assert got[:, 2:].tobytes() == want[:, 2:].tobytes()


@pytest.mark.parametrize("regime", ["ring", "ring_noise_1cm"])
def test_exchange_slot_written_by_the_kernel_equals_the_packed_arena(gpu_mode, regime):
    """rplgpu_cloud_arena_xyi_dev (the voxel kernel writes the 12-byte exchange points itself, incl.
    the multi-band scans that go through the temporary cell area) against rplgpu_cloud_arena_dev:
    per scan the same (x, y, intensity), bit for bit; a slot too small cuts and flags, never overruns."""
    import torch
    gpu = gpu_mode
    dev = torch.device("cuda:0")
    B, n = 192, 32000
    batch = synth.make_batch(2027, B, n, **({"noise_m": 0.01} if regime != "ring" else {}))
    p = Params.defaults(clip_enable=1, q_min=0, range_min=0.15, range_max=40.0, voxel_enable=1,
                        voxel_leaf=0.05)
    arena, start, npts, st, total, _ = _run_arena(gpu, batch, p, n, with_keys=False)
    assert int(st.max()) == 0
    d_nodes = torch.from_numpy(batch.view(np.uint8).reshape(B, n * 8)).to(dev)
    d_len = torch.full((B,), n, dtype=torch.int32, device=dev)
    for slot in (B * n, total, total - 1000):
        d_slot = torch.full((slot + 64, 3), -7.0, dtype=torch.float32, device=dev)  # (+ guard rows)
        d_cur = torch.zeros(1, dtype=torch.int64, device=dev)
        d_start = torch.zeros(B, dtype=torch.int64, device=dev)
        d_np = torch.zeros(B, dtype=torch.int32, device=dev)
        d_st = torch.zeros(B, dtype=torch.int32, device=dev)
        gpu.cloud_arena_xyi_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, p, d_slot.data_ptr(), slot,
                                d_cur.data_ptr(), d_start.data_ptr(), d_np.data_ptr(), d_st.data_ptr())
        gpu.synchronize()
        pts, s2, n2, st2 = d_slot.cpu().numpy(), d_start.cpu().numpy(), d_np.cpu().numpy().astype(np.int64), d_st.cpu().numpy()
        assert np.all(pts[slot:] == -7.0)  # nothing behind the slot
        assert int(d_cur.item()) == total  # (the cursor counts every cell, kept or cut)
        if slot >= total:
            assert int(st2.max()) == 0 and np.array_equal(n2, npts)
        else:
            assert int(n2.sum()) <= slot and np.any(st2 & 8)  # RPLGPU_SCAN_OUT_TRUNCATED
            assert np.all((n2 == npts) | ((st2 & 8) != 0))
        for b in range(B):
            if n2[b] == npts[b] and npts[b]:
                want = arena[start[b]: start[b] + npts[b]][:, [0, 1, 3]]
                assert pts[s2[b]: s2[b] + n2[b]].tobytes() == np.ascontiguousarray(want).tobytes(), (slot, b)


def test_runs_across_dropped_lanes_patterns(gpu_mode, oracle):
    """The FILL passes (a class of a two-class block, the quality filter, the E5 mask) let a run cross dropped
    samples and up to two wholly dropped lanes (a lane = two consecutive samples).  Scans built to sit on every
    branch of that: points alternating between two or three cells in groups of 1 ... 7 samples (so that the other
    class's — or the low-quality — samples fill 0, 1, 2, 3 whole lanes between two samples of one cell), at every
    lane phase and across the 128-sample block ends and the lane 63 | lane 0 seam, with random invalid samples on
    top; with and without the quality filter and the ROR mask.  Every scan against the oracle, in every form."""
    gpu = gpu_mode
    rng = np.random.default_rng(515)
    B, n = 96, 4096
    batch = np.zeros((B, n), synth.NODE_DTYPE)
    for b in range(B):
        group = 1 + b % 7                      # samples per stretch in one cell
        ncell = 2 + (b // 7) % 2               # cells the stretches rotate through
        phase = int(rng.integers(0, 2 * group + 1))
        i = np.arange(n)
        which = ((i + phase) // group) % ncell
        ang = (i.astype(np.float64) / n) * 2.0 * np.pi
        r0 = 3.0 + 0.37 * (b % 11)
        # radial offsets that put neighbouring stretches into different 5 cm cells (and, for odd b, into cells
        # of the SAME checkerboard colour two cells apart as well)
        step = 0.05 if b % 2 == 0 else 0.10
        r = r0 + which * step + rng.normal(0.0, 0.002, n)
        d = np.clip(np.round(r * 4000.0), 1, 2 ** 31).astype(np.uint32)
        q = rng.integers(0, 256, n).astype(np.uint8) if b % 3 == 0 else np.where(which == 0, 200, 20 + (b % 5) * 10).astype(np.uint8)
        inv = rng.random(n) < (0.0 if b % 4 else 0.06)
        d[inv] = 0
        batch[b]["angle_z_q14"] = np.round(ang * (32768.0 / np.pi)).astype(np.int64) % 65536
        batch[b]["dist_mm_q2"] = d
        batch[b]["quality"] = q
        batch[b]["flag"] = 0
    for qmin, ror in ((0, 0), (48, 0), (120, 0), (0, 1), (48, 1)):
        p = Params.defaults(clip_enable=1, q_min=qmin, range_min=0.15, range_max=40.0, voxel_enable=1,
                            voxel_leaf=0.05, ror_enable=ror, ror_radius=0.08, ror_min_neighbors=2)
        _check_batch(gpu, oracle, batch, p, cap_per_scan=n)


# --------------------------------------------------------------------------------------------
# The PINNED rows at the bench scale (missing #2): every scan of the batches that
# bench.py's `reference_path_gpu` times — the headline batch (seed 2026, exactly uniform angle words)
# and the jitter regimes (seed 2026 + 11, +-1 / 3 / 10 / 20 / 64 / 300 words) — through the batch entry points of
# ascendScanData (src/sdk/src/sl_lidar_driver.cpp:128-184) and publish_scan
# (src/rplidar_node.cpp:583-680), compared scan by scan with the oracle on all host cores
# (oracle/oracle.cpp orc_batch_ascend_check / orc_batch_laserscan_check; the oracle itself is pinned
# to the reference compiled here, tests/test_oracle_golden.py).
# --------------------------------------------------------------------------------------------
_BENCH_REGIMES = {"uniform_angles": (2026, 0), "jitter1": (2037, 1), "jitter3": (2037, 3),
                  "jitter10": (2037, 10), "jitter20": (2037, 20), "jitter64": (2037, 64), "jitter300": (2037, 300)}


@pytest.fixture(scope="module", params=list(_BENCH_REGIMES))
def bench_regime(request):
    """(name, batch) of one bench regime; module-scoped so that the tests below share the 1 GB batch
    (pytest runs all tests of one parameter before it builds the next)."""
    seed, jit = _BENCH_REGIMES[request.param]
    return request.param, synth.make_batch(seed, 4096, 32000, **({"jitter": jit} if jit else {}))


def _ascend_sorted(gpu):
    from rplidar_ros2_driver_amd import abi
    cnt = C.c_uint32(0)
    lib = abi.load_library()
    lib.rplgpu_debug_ascend_sorted.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    assert lib.rplgpu_debug_ascend_sorted(gpu._h, C.byref(cnt)) == 0
    return int(cnt.value)


def _assert_ascend(oracle, src, got, lens, tag):
    bad, res = oracle.batch_ascend_check(src, got, lens, os.cpu_count() or 1)
    first = np.nonzero(res[:, 1:].any(axis=1))[0]
    assert bad == 0, (tag, bad, first[:8], res[first[:8]])
    return res


def test_bench_batches_ascend_matches_oracle(gpu, oracle, bench_regime):
    """rplgpu_ascend_batch_dev over all 4096 scans of a bench regime: the oracle's angle words at
    every position, the oracle's nodes after canonicalising equal-angle runs, the valid nodes in
    stable order (this library's tie rule), the slot tails untouched, status 0 — in the configuration
    bench.py times; +-64 words exercises the chunk merges of round 6 on every boundary, +-300 words the
    sorting kernel (k_ascend<true>) on (nearly) every scan."""
    import torch
    dev = torch.device("cuda:0")
    regime, batch = bench_regime
    B, n = batch.shape
    lens = np.full(B, n, np.uint32)
    d_nodes = torch.from_numpy(batch.view(np.uint8).reshape(B, n * 8)).to(dev)
    d_len = torch.from_numpy(lens.astype(np.int32)).to(dev)
    d_st = torch.full((B,), -1, dtype=torch.int32, device=dev)
    gpu.ascend_batch_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, d_st.data_ptr())
    gpu.synchronize()
    nsorted = _ascend_sorted(gpu)
    assert int(d_st.abs().max()) == 0
    got = d_nodes.cpu().numpy().view(synth.NODE_DTYPE).reshape(B, n)
    res = _assert_ascend(oracle, batch, got, lens, regime)
    assert not res[:, 0].any()  # every scan has valid samples: SL_RESULT_OK
    # which kernel did the work (round 6: chunk merges keep everything up to +-128 words — nodes up to 128 places
    # from home — in the streaming kernel; +-300 words is the sorting kernel's, k_ascend<true>)
    if regime == "jitter300":
        assert nsorted > B // 2, nsorted
    else:
        assert nsorted == 0, nsorted
    # a second batch of another kind for free: the ascended scans ascended again (their invalid
    # nodes now sit between the valid ones and are given the angle of their NEW place, so this is not
    # an identity beyond +-1) — again against the oracle, scan by scan
    d_again = d_nodes.clone()
    gpu.ascend_batch_dev(d_again.data_ptr(), n, d_len.data_ptr(), B, d_st.data_ptr())
    gpu.synchronize()
    assert int(d_st.abs().max()) == 0
    again = d_again.cpu().numpy().view(synth.NODE_DTYPE).reshape(B, n)
    _assert_ascend(oracle, got, again, lens, (regime, "again"))
    if regime in ("uniform_angles", "jitter1"):
        assert torch.equal(d_again, d_nodes)


def test_bench_batches_laserscan_matches_oracle(gpu, oracle, bench_regime):
    """rplgpu_laserscan_batch_dev (Mode A and Mode B, inverted on / off) and
    rplgpu_ascend_laserscan_batch_dev (S1 -> S3 in one pass, with and without the ascended nodes)
    over all 4096 scans of a bench regime: beam counts, ranges and intensities word for word
    against orc_publish_scan, the metadata of every scan (rplgpu_fill_meta) against the oracle's."""
    import torch
    dev = torch.device("cuda:0")
    regime, batch = bench_regime
    B, n = batch.shape
    lens = np.full(B, n, np.uint32)
    d_nodes = torch.from_numpy(batch.view(np.uint8).reshape(B, n * 8)).to(dev)
    d_len = torch.from_numpy(lens.astype(np.int32)).to(dev)
    d_r = torch.empty(B, n, dtype=torch.float32, device=dev)
    d_i = torch.empty(B, n, dtype=torch.float32, device=dev)
    d_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    threads = os.cpu_count() or 1

    def check(p, tag):
        cnt = d_cnt.cpu().numpy().astype(np.uint32)
        bad, res = oracle.batch_laserscan_check(batch, lens, oracle_lib.copy_params(p), d_r.cpu().numpy(),
                                                d_i.cpu().numpy(), cnt, threads)
        first = np.nonzero(res[:, 1:].any(axis=1))[0]
        assert bad == 0, (regime, tag, bad, first[:8], res[first[:8]])
        assert np.array_equal(res[:, 0], cnt)
        return cnt

    for sp in (1, 0):
        for inv in (0, 1):
            p = Params.defaults(range_max=40.0, scan_processing=sp, inverted=inv)
            d_r.fill_(-3.0)
            d_i.fill_(-3.0)
            gpu.laserscan_batch_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, p, d_r.data_ptr(),
                                    d_i.data_ptr(), d_cnt.data_ptr())
            gpu.synchronize()
            cnt = check(p, ("laserscan", sp, inv))
            # nothing behind a scan's beams is written
            tail = torch.arange(n, device=dev)[None, :] >= d_cnt[:, None]
            assert bool((d_r[tail] == -3.0).all()) and bool((d_i[tail] == -3.0).all())
            # the metadata scalars of a few scans: the library's against the oracle's, byte for byte
            for b in (0, 1, B // 2, B - 1):
                m = gpu.fill_meta(p, int(cnt[b]), 0.1)
                _, _, wm = oracle.publish_scan(batch[b], oracle_lib.copy_params(p), 0.1)
                assert bytes(m) == bytes(wm), (regime, sp, inv, b)
    # S1 -> S3 in one pass: the nodes only read ...
    p = Params.defaults(range_max=40.0)
    before = d_nodes.clone()
    d_r.fill_(-3.0)
    d_i.fill_(-3.0)
    gpu.ascend_laserscan_batch_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, p, d_r.data_ptr(),
                                   d_i.data_ptr(), d_cnt.data_ptr())
    gpu.synchronize()
    assert torch.equal(before, d_nodes)
    check(p, "one_pass")
    # ... and the form that leaves the ascended nodes as well
    d_st = torch.full((B,), -1, dtype=torch.int32, device=dev)
    gpu.ascend_laserscan_batch_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, p, d_r.data_ptr(),
                                   d_i.data_ptr(), d_cnt.data_ptr(), True, d_st.data_ptr())
    gpu.synchronize()
    assert int(d_st.abs().max()) == 0
    check(p, "one_pass_with_nodes")
    got = d_nodes.cpu().numpy().view(synth.NODE_DTYPE).reshape(B, n)
    _assert_ascend(oracle, batch, got, lens, (regime, "one_pass_with_nodes"))


def test_voxel_aggregation_auto_converges_per_batch(gpu, oracle):
    """RPLGPU_VOXEL_AGG_AUTO picks the block-aggregation instance of a batch from THAT batch's own
    queue statistics (its previous launch).  ADVICE r5: with one remembered batch identity a caller
    that alternates two staging buffers, or launches a batch in chunks (the exchange pipeline does),
    never matched and stayed on the plain instance for noisy data.  Decisions are now kept for the last
    eight identities: a noisy and a clean batch launched alternately each settle on their own instance
    from the second launch on, and so do the two halves of a noisy batch launched as chunks."""
    import torch
    from rplidar_ros2_driver_amd import abi
    lib = abi.load_library()
    lib.rplgpu_debug_voxel_instance.argtypes = [C.c_void_p]
    dev = torch.device("cuda:0")
    B, n = 256, 32000
    noisy = synth.make_batch(2026, B, n, noise_m=0.01)
    clean = synth.make_batch(2026, B, n)
    p = Params.defaults(clip_enable=1, q_min=0, range_min=0.15, range_max=40.0, voxel_enable=1, voxel_leaf=0.05)
    d = {k: torch.from_numpy(v.view(np.uint8).reshape(B, n * 8)).to(dev) for k, v in (("noisy", noisy), ("clean", clean))}
    d_len = torch.full((B,), n, dtype=torch.int32, device=dev)
    cap = B * n
    d_arena = torch.empty(cap, 4, dtype=torch.float32, device=dev)
    d_cur = torch.zeros(1, dtype=torch.int64, device=dev)
    d_start = torch.zeros(B, dtype=torch.int64, device=dev)
    d_np = torch.zeros(B, dtype=torch.int32, device=dev)
    d_st = torch.zeros(B, dtype=torch.int32, device=dev)
    gpu.set_voxel_aggregation(0)

    def launch(buf, lo, cnt):
        gpu.cloud_arena_dev(buf.data_ptr() + lo * n * 8, n, d_len.data_ptr(), cnt, p, d_arena.data_ptr(), cap,
                            d_cur.data_ptr(), d_start.data_ptr(), d_np.data_ptr(), d_st.data_ptr())
        inst = lib.rplgpu_debug_voxel_instance(gpu._h)
        gpu.synchronize()  # (the statistics of this launch are in pinned memory before the next decision)
        assert int(d_st.max()) == 0
        return inst, int(d_cur.item())

    seen = {"noisy": [], "clean": []}
    cells = {"noisy": set(), "clean": set()}
    for rnd in range(4):
        for k in ("noisy", "clean"):
            inst, total = launch(d[k], 0, B)
            seen[k].append(inst)
            cells[k].add(total)
    assert seen["noisy"] == [0, 1, 1, 1], seen
    assert seen["clean"] == [0, 0, 0, 0], seen
    assert len(cells["noisy"]) == 1 and len(cells["clean"]) == 1  # either instance makes the same cloud
    # a noisy batch in two chunks (two identities), three steps
    halves = []
    for step in range(3):
        halves.append([launch(d["noisy"], lo, B // 2)[0] for lo in (0, B // 2)])
    assert halves == [[0, 0], [1, 1], [1, 1]], halves
    # the whole cloud against the oracle once, in the instance AUTO settled on
    launch(d["noisy"], 0, B)
    arena, start, npts = d_arena.cpu().numpy(), d_start.cpu().numpy(), d_np.cpu().numpy().astype(np.int64)
    bad, res = oracle.batch_cloud_check(noisy, oracle_lib.copy_params(p), arena, start, npts, None, os.cpu_count() or 1)
    assert bad == 0
    assert float(res[:, 3].copy().view(np.float32).max()) <= XYZ_TOL
