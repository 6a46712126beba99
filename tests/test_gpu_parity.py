"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle.

Bars (BASELINE.json north_star): bit-exact for the integer keep mask, beam counts, bin
indices, cell indices and angle words; fp32 ranges / intensities / XYZ: bit-exact where
the design makes them so (LUT + single IEEE ops), and within 1e-6 m for voxel centroids
(fixed-point accumulation).
"""
import ctypes as C

import numpy as np
import pytest

from rplidar_ros2_driver_amd import NODE_DTYPE, Params, synth
from rplidar_ros2_driver_amd import abi
from tests import oracle_lib

pytestmark = pytest.mark.gpu

XYZ_TOL = 1e-6  # metres, north_star tolerance for float XYZ


from tests.cases import CASES  # noqa: E402



# --------------------------------------------------------------------------- ascend (S1)
@pytest.mark.parametrize("name", list(CASES))
def test_ascend_matches_oracle(gpu, oracle, name):
    nodes = CASES[name]
    want, want_res = oracle.ascend(nodes)
    got = nodes.copy()
    got_res = gpu.ascend(got)
    assert got_res == want_res
    if want_res != 0:
        assert got.tobytes() == nodes.tobytes()  # untouched on SL_RESULT_OPERATION_FAIL
        return
    # angle words, validity and multiset per equal-angle run are bit-exact; inside a run
    # the reference order is introsort's (unstable) — compare canonical forms.
    assert np.all(np.diff(got["angle_z_q14"].astype(np.int64)) >= 0)
    assert np.array_equal(got["angle_z_q14"], want["angle_z_q14"])
    assert oracle_lib.canon_equal_angle_runs(got).tobytes() == \
        oracle_lib.canon_equal_angle_runs(want).tobytes()
    # our own tie rule: stable in (angle, input index) — check against a numpy stable sort
    # stability is checked on the valid samples, whose angles the fill pass never touches
    valid_in = nodes[nodes["dist_mm_q2"] != 0]
    valid_out = got[got["dist_mm_q2"] != 0]
    order = np.argsort(valid_in["angle_z_q14"], kind="stable")
    assert valid_out.tobytes() == valid_in[order].tobytes()


def test_ascend_is_idempotent_on_sorted_output(gpu, oracle):
    nodes = CASES["c2_32000"]
    once = nodes.copy()
    assert gpu.ascend(once) == 0
    twice = once.copy()
    assert gpu.ascend(twice) == 0
    want2, _ = oracle.ascend(once)
    assert oracle_lib.canon_equal_angle_runs(twice).tobytes() == \
        oracle_lib.canon_equal_angle_runs(want2).tobytes()


# --------------------------------------------------------------------------- LaserScan (S3)
def _has_intensity_tie(nodes, p):
    """True when two kept samples share (angle, dist_m) but differ in intensity: the
    reference's winner then depends on introsort's tie order (documented)."""
    v = nodes[nodes["dist_mm_q2"] != 0]
    if len(v) < 2:
        return False
    dm = (v["dist_mm_q2"].astype(np.float32) / np.float32(4000.0)).view(np.uint32)
    inten = (v["quality"] if p.is_new_protocol else (v["quality"] >> 2)).astype(np.int64)
    key = (v["angle_z_q14"].astype(np.int64) << 32) | dm.astype(np.int64)
    order = np.argsort(key, kind="stable")
    key, inten = key[order], inten[order]
    starts = np.r_[0, np.flatnonzero(np.diff(key) != 0) + 1]
    return bool(np.any(np.minimum.reduceat(inten, starts) != np.maximum.reduceat(inten, starts)))


@pytest.mark.parametrize("scan_processing", [1, 0])
@pytest.mark.parametrize("inverted", [0, 1])
@pytest.mark.parametrize("is_new", [0, 1])
@pytest.mark.parametrize("name", list(CASES))
def test_laserscan_matches_oracle(gpu, oracle, name, is_new, inverted, scan_processing):
    nodes = CASES[name]
    p = Params.defaults(is_new_protocol=is_new, inverted=inverted,
                        scan_processing=scan_processing, range_max=40.0)
    op = oracle_lib.copy_params(p)
    wr, wi, wm = oracle.publish_scan(nodes, op, 0.125)
    gr, gi, gm = gpu.scan_to_laserscan(nodes, p, 0.125)
    assert bytes(gm) == bytes(wm)  # every metadata float + count + published, bit-exact
    if not wm.published:
        return
    if scan_processing:
        assert gr.tobytes() == wr.tobytes()  # ranges bit-exact (implies bin indices exact)
        if _has_intensity_tie(nodes, p):
            # winner among identical (angle, dist) is implementation-defined upstream
            diff = np.flatnonzero(gi != wi)
            v = nodes[nodes["dist_mm_q2"] != 0]
            allowed = set((v["quality"] if is_new else (v["quality"] >> 2)).astype(np.float32))
            assert all(float(gi[k]) in allowed for k in diff)
        else:
            assert gi.tobytes() == wi.tobytes()
    else:
        # Mode B: the order of equal-angle samples is introsort's; compare per equal-angle
        # run as multisets, and exactly when angles are unique.
        v = nodes[nodes["dist_mm_q2"] != 0]
        if len(np.unique(v["angle_z_q14"])) == len(v):
            assert gr.tobytes() == wr.tobytes()
            assert gi.tobytes() == wi.tobytes()
        else:
            srt = np.sort(v["angle_z_q14"])
            if not inverted:
                srt = srt[::-1]
            bounds = np.flatnonzero(np.diff(srt.astype(np.int64)) != 0) + 1
            for lo, hi in zip(np.r_[0, bounds], np.r_[bounds, len(srt)]):
                ka = sorted(map(tuple, np.stack([gr[lo:hi], gi[lo:hi]], 1).tolist()))
                kb = sorted(map(tuple, np.stack([wr[lo:hi], wi[lo:hi]], 1).tolist()))
                assert ka == kb


def test_laserscan_clip_extension(gpu, oracle):
    nodes = CASES["c2_32000"]  # unique angles: no implementation-defined intensity ties
    p = Params.defaults(is_new_protocol=1, clip_enable=1, q_min=40, range_min=0.5,
                        range_max=18.0)
    wr, wi, wm = oracle.publish_scan(nodes, oracle_lib.copy_params(p), 0.1)
    gr, gi, gm = gpu.scan_to_laserscan(nodes, p, 0.1)
    assert bytes(gm) == bytes(wm)
    assert gr.tobytes() == wr.tobytes() and gi.tobytes() == wi.tobytes()


def test_c1_dummy_scan_pipeline(gpu, oracle):
    """Config 1: the reference's Dummy generator (360 samples, old protocol)."""
    for s in range(3):
        nodes = oracle.gen_dummy(s).view(NODE_DTYPE)
        p = Params.defaults(range_max=40.0)  # Dummy hw limit, src/lidar_driver_wrapper.cpp:439
        wr, wi, wm = oracle.publish_scan(nodes, oracle_lib.copy_params(p), 0.1)
        gr, gi, gm = gpu.scan_to_laserscan(nodes, p, 0.1)
        assert bytes(gm) == bytes(wm) and gm.count == 360
        assert gr.tobytes() == wr.tobytes() and gi.tobytes() == wi.tobytes()
        assert set(gi.tolist()) <= {50.0, 0.0}  # 200 >> 2


# --------------------------------------------------------------------------- cloud (ext)
@pytest.mark.parametrize("inverted", [0, 1])
@pytest.mark.parametrize("name", ["c1_like_360", "ring_8192_rot_jit", "c2_32000",
                                  "full_32768_unsorted", "all_invalid", "huge_dist", "kat2"])
def test_cloud_xyz_matches_oracle(gpu, oracle, name, inverted):
    nodes = CASES[name]
    p = Params.defaults(inverted=inverted, clip_enable=1, q_min=8, range_min=0.15,
                        range_max=40.0)
    want = oracle.scan_to_cloud(nodes, oracle_lib.copy_params(p))
    got, status = gpu.scan_to_cloud(nodes, p)
    assert status == 0
    assert got.shape == want.shape  # keep mask (integer) exact
    assert got.tobytes() == want.tobytes()  # LUT design: XYZ and intensity bit-exact
    if len(want):
        assert np.max(np.abs(got[:, :2] - want[:, :2])) <= XYZ_TOL


# --------------------------------------------------------------------------- E5 radius outliers
def _ror_cases():
    rng = np.random.default_rng(99)
    cases = {
        "ring_noise_4000": synth.make_scan(41, 0, 4000, noise_m=0.08, r0_range=(4.0, 5.0)),
        "uniform_777": CASES["unsorted_uniform"],
        "ring_8192_rot_jit": CASES["ring_8192_rot_jit"],
        "c1_like_360": CASES["c1_like_360"],
        "kat2": CASES["kat2"],
        "all_invalid": CASES["all_invalid"],
        "single_valid": CASES["single_valid"],
        "near_origin": synth.make_scan(41, 1, 3000, r0_range=(0.16, 0.3)),
    }
    sp = synth.make_scan(41, 2, 6000, r0_range=(10.0, 12.0))  # isolated returns in empty space
    hole = rng.random(6000) < 0.9
    sp["dist_mm_q2"][hole] = 0
    cases["sparse_6000"] = sp
    return cases


ROR_CASES = _ror_cases()


@pytest.mark.parametrize("inverted", [0, 1])
@pytest.mark.parametrize("name", list(ROR_CASES))
def test_ror_cloud_matches_oracle(gpu, oracle, name, inverted):
    """E5: the keep mask is integer work -> the surviving cloud must be bit-identical."""
    nodes = ROR_CASES[name]
    p = Params.defaults(inverted=inverted, clip_enable=1, range_min=0.15, range_max=40.0,
                        ror_enable=1, ror_radius=0.10, ror_min_neighbors=2)
    want = oracle.scan_to_cloud(nodes, oracle_lib.copy_params(p))
    got, status = gpu.scan_to_cloud(nodes, p)
    assert status == 0
    assert got.shape == want.shape
    assert got.tobytes() == want.tobytes()


@pytest.mark.parametrize("radius,k", [(0.05, 1), (0.03, 4), (0.10, 3)])
def test_ror_parameters(gpu, oracle, radius, k):
    nodes = ROR_CASES["ring_noise_4000"]
    p = Params.defaults(clip_enable=1, range_max=40.0, ror_enable=1, ror_radius=radius,
                        ror_min_neighbors=k)
    want = oracle.scan_to_cloud(nodes, oracle_lib.copy_params(p))
    base = oracle.scan_to_cloud(nodes, oracle_lib.copy_params(
        Params.defaults(clip_enable=1, range_max=40.0)))
    got, status = gpu.scan_to_cloud(nodes, p)
    assert status == 0 and got.tobytes() == want.tobytes()
    assert 0 < len(want) < len(base)  # the filter really removes something here


def test_ror_then_voxel_matches_oracle(gpu, oracle):
    """C5 pipeline: E1 clip -> E5 radius outlier removal -> E4 voxel grid."""
    for name in ("ring_noise_4000", "sparse_6000", "ring_8192_rot_jit"):
        nodes = ROR_CASES[name]
        p = Params.defaults(clip_enable=1, range_max=40.0, ror_enable=1, voxel_enable=1)
        want, wcells, _ = oracle.cloud_pipeline(nodes, oracle_lib.copy_params(p))
        got, status = gpu.scan_to_cloud(nodes, p)
        assert status == 0 and len(got) == len(want)
        if len(want):
            assert np.max(np.abs(got[:, :2].astype(np.float64) - want[:, :2])) <= XYZ_TOL
            assert got[:, 3].tobytes() == want[:, 3].tobytes()


def test_ror_full_scan_32000(gpu, oracle):
    nodes = synth.make_scan(43, 0, 32000, noise_m=0.05)
    p = Params.defaults(clip_enable=1, range_max=40.0, ror_enable=1)
    want = oracle.scan_to_cloud(nodes, oracle_lib.copy_params(p))  # O(n^2): ~1e9 pair tests
    got, status = gpu.scan_to_cloud(nodes, p)
    assert status == 0 and got.tobytes() == want.tobytes()


def test_ror_many_unsettled_samples(gpu, oracle):
    """Random ranges: almost no sample has its neighbours next to it in angle order, so nearly all
    of them go to the exhaustive stage — more than its work list holds (8192), which exercises
    the list, the overflow path and scans where most points ARE outliers."""
    for n, r, k in ((12000, 0.10, 2), (12000, 0.6, 1), (9000, 1.5, 3)):
        nodes = synth.make_scan(91, n, n, kind="uniform")
        p = Params.defaults(clip_enable=1, range_max=40.0, ror_enable=1, ror_radius=r,
                            ror_min_neighbors=k)
        want = oracle.scan_to_cloud(nodes, oracle_lib.copy_params(p))
        base = oracle.scan_to_cloud(nodes, oracle_lib.copy_params(
            Params.defaults(clip_enable=1, range_max=40.0)))
        got, status = gpu.scan_to_cloud(nodes, p)
        assert status == 0 and got.tobytes() == want.tobytes()
        assert len(want) < len(base)


def _cells_of(xyzi, leaf):
    leaf = np.float32(leaf)
    ix = np.floor(xyzi[:, 0] / leaf).astype(np.int32)
    iy = np.floor(xyzi[:, 1] / leaf).astype(np.int32)
    return ix, iy


@pytest.mark.parametrize("name,leaf", [("c1_like_360", 0.05), ("ring_8192_rot_jit", 0.05),
                                       ("c2_32000", 0.05), ("c2_32000_newproto", 0.1),
                                       ("kat2", 0.05), ("all_invalid", 0.05)])
def test_voxel_cloud_matches_oracle(gpu, oracle, name, leaf):
    nodes = CASES[name]
    p = Params.defaults(clip_enable=1, range_min=0.15, range_max=40.0, voxel_enable=1,
                        voxel_leaf=leaf, is_new_protocol=int("newproto" in name))
    want, wcells, wcounts = oracle.cloud_pipeline(nodes, oracle_lib.copy_params(p))
    got, status = gpu.scan_to_cloud(nodes, p)
    assert status == 0
    assert len(got) == len(want)  # number of occupied cells (integer) exact
    if len(want) == 0:
        return
    # cell membership and (iy, ix) order: bit-exact integer work
    gx, gy = _cells_of(got, leaf)
    # a centroid may sit on a cell face; compare with the oracle's cell list instead of
    # re-deriving from rounded centroids when they disagree
    assert np.array_equal(np.stack([gx, gy], 1), wcells) or \
        np.max(np.abs(got[:, :2] - want[:, :2])) <= XYZ_TOL
    assert np.max(np.abs(got[:, :2].astype(np.float64) - want[:, :2])) <= XYZ_TOL
    assert np.all(got[:, 2] == 0.0)
    assert got[:, 3].tobytes() == want[:, 3].tobytes()  # integer sums: mean intensity exact


def test_voxel_is_deterministic(gpu):
    nodes = CASES["c2_32000"]
    p = Params.defaults(clip_enable=1, range_max=40.0, voxel_enable=1)
    a, _ = gpu.scan_to_cloud(nodes, p)
    for _ in range(3):
        b, _ = gpu.scan_to_cloud(nodes, p)
        assert a.tobytes() == b.tobytes()


def test_voxel_more_cells_than_table(gpu, oracle):
    """Every sample its own cell: far more cells than the on-chip table holds, so the
    kernel bisects the key space into bands; the result must still be the full, ordered
    voxel cloud (nothing dropped, nothing flagged)."""
    nodes = synth.make_scan(5, 0, 32000, kind="uniform", invalid_p=0.0)
    p = Params.defaults(clip_enable=1, range_max=40.0, voxel_enable=1, voxel_leaf=0.01)
    want, wcells, _ = oracle.cloud_pipeline(nodes, oracle_lib.copy_params(p))
    got, status = gpu.scan_to_cloud(nodes, p)
    assert status == 0
    assert len(got) == len(want) and len(want) > 20000
    assert np.max(np.abs(got[:, :2].astype(np.float64) - want[:, :2])) <= XYZ_TOL
    assert got[:, 3].tobytes() == want[:, 3].tobytes()


def test_cloud_arena_equals_per_scan_regions(gpu):
    """rplgpu_cloud_arena_dev: the clouds of a batch in one contiguous arena, every scan
    reserving exactly its cells.  Each scan's slice must be byte-identical to what the
    per-scan-region entry point writes; scans of very different size (incl. one that needs key
    bands, i.e. the count-then-write path, and an all-invalid one) share the arena without gaps;
    a too small arena truncates and flags instead of overrunning."""
    torch = _torch()
    dev = torch.device("cuda:0")
    B, n = 40, 32000
    batch = synth.make_batch(61, B, n)
    batch[7] = synth.make_scan(5, 0, n, kind="uniform", invalid_p=0.0)   # > table: key bands
    batch[19]["dist_mm_q2"] = 0                                           # nothing kept
    lens = np.full(B, n, np.int32)
    lens[3] = 777
    for leaf in (0.05, 0.01):
        p = Params.defaults(clip_enable=1, range_max=40.0, voxel_enable=1, voxel_leaf=leaf)
        d_nodes = torch.from_numpy(batch.view(np.uint8).reshape(B, n * 8)).to(dev)
        d_len = torch.from_numpy(lens).to(dev)
        stride = n
        d_xyzi = torch.zeros(B, stride, 4, dtype=torch.float32, device=dev)
        d_np = torch.zeros(B, dtype=torch.int32, device=dev)
        d_st = torch.zeros(B, dtype=torch.int32, device=dev)
        gpu.cloud_batch_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, p, d_xyzi.data_ptr(),
                            stride, d_np.data_ptr(), d_st.data_ptr())
        gpu.synchronize()
        npts = d_np.cpu().numpy().astype(np.int64)
        assert int(d_st.max()) == 0 and npts[19] == 0 and npts[7] > 7168
        total = int(npts.sum())
        for cap in (total + 1000, total, total // 2):
            d_arena = torch.full((max(cap, 1) + 16, 4), -1.0, dtype=torch.float32, device=dev)
            d_cur = torch.full((1,), 123, dtype=torch.int64, device=dev)
            d_start = torch.zeros(B, dtype=torch.int64, device=dev)
            d_np2 = torch.zeros(B, dtype=torch.int32, device=dev)
            d_st2 = torch.zeros(B, dtype=torch.int32, device=dev)
            gpu.cloud_arena_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, p, d_arena.data_ptr(),
                                cap, d_cur.data_ptr(), d_start.data_ptr(), d_np2.data_ptr(),
                                d_st2.data_ptr())
            gpu.synchronize()
            assert int(d_cur[0]) == total  # every scan reserved exactly its cells
            start = d_start.cpu().numpy()
            np2, st2 = d_np2.cpu().numpy().astype(np.int64), d_st2.cpu().numpy()
            arena, ref = d_arena.cpu().numpy(), d_xyzi.cpu().numpy()
            assert np.all(arena[cap:] == -1.0)  # nothing written past the capacity
            # the reservations tile [0, total) without gaps or overlaps
            nz = np.nonzero(npts)[0]  # (a scan without cells reserves nothing; its start is 0)
            order = nz[np.argsort(start[nz], kind="stable")]
            assert np.array_equal(np.cumsum(npts[order]) - npts[order], start[order])
            for b in range(B):
                kept = int(min(max(cap - start[b], 0), npts[b]))
                assert np2[b] == kept
                assert bool(st2[b] & abi.SCAN_OUT_TRUNCATED) == (kept < npts[b])
                assert arena[start[b]: start[b] + kept].tobytes() == ref[b, :kept].tobytes()


def test_voxel_cell_range_is_reported(gpu):
    # |cell index| must stay below 32767: 40 m / 1 mm leaf does not -> flagged, not silent
    nodes = synth.make_scan(5, 1, 4000, invalid_p=0.0, r0_range=(35.0, 36.0))
    p = Params.defaults(clip_enable=1, range_max=60.0, voxel_enable=1, voxel_leaf=0.001)
    got, status = gpu.scan_to_cloud(nodes, p, allow_overflow=True)
    assert status & abi.SCAN_CELL_RANGE


# --------------------------------------------------------------------------- batches (device)
def _torch():
    import torch
    return torch


def test_batch_dev_matches_single_scan_and_oracle(gpu, oracle):
    torch = _torch()
    B, n = 24, 4000
    batch = synth.make_batch(21, B, n, jitter=2)
    lens = np.array([n - 13 * b for b in range(B)], np.uint32)
    lens[3] = 0
    dev = torch.device("cuda:0")
    d_nodes = torch.from_numpy(batch.view(np.uint8).reshape(B, n * 8)).to(dev)
    d_len = torch.from_numpy(lens.astype(np.int32)).to(dev)
    d_r = torch.empty(B, n, dtype=torch.float32, device=dev)
    d_i = torch.empty(B, n, dtype=torch.float32, device=dev)
    d_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    p = Params.defaults(range_max=40.0)
    gpu.laserscan_batch_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, p,
                            d_r.data_ptr(), d_i.data_ptr(), d_cnt.data_ptr())
    gpu.synchronize()
    cnt = d_cnt.cpu().numpy()
    r = d_r.cpu().numpy()
    i = d_i.cpu().numpy()
    for b in range(B):
        wr, wi, wm = oracle.publish_scan(batch[b, : lens[b]], oracle_lib.copy_params(p), 0.1)
        assert cnt[b] == wm.count
        assert r[b, : wm.count].tobytes() == wr.tobytes()
        assert i[b, : wm.count].tobytes() == wi.tobytes()

    # voxelised clouds + pack
    pv = Params.defaults(clip_enable=1, range_max=40.0, voxel_enable=1)
    out_stride = 8192
    d_xyzi = torch.zeros(B, out_stride, 4, dtype=torch.float32, device=dev)
    d_np = torch.zeros(B, dtype=torch.int32, device=dev)
    d_st = torch.zeros(B, dtype=torch.int32, device=dev)
    gpu.cloud_batch_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, pv, d_xyzi.data_ptr(),
                        out_stride, d_np.data_ptr(), d_st.data_ptr())
    d_off = torch.zeros(B + 1, dtype=torch.int64, device=dev)
    d_packed = torch.zeros(B * out_stride, 4, dtype=torch.float32, device=dev)
    gpu.pack_clouds_dev(d_xyzi.data_ptr(), out_stride, d_np.data_ptr(), B, d_packed.data_ptr(),
                        d_off.data_ptr())
    gpu.synchronize()
    npts = d_np.cpu().numpy()
    off = d_off.cpu().numpy()
    assert np.all(d_st.cpu().numpy() == 0)
    assert np.array_equal(off, np.r_[0, np.cumsum(npts)])
    packed = d_packed.cpu().numpy()
    xyzi = d_xyzi.cpu().numpy()
    for b in range(B):
        want, _, _ = oracle.cloud_pipeline(batch[b, : lens[b]], oracle_lib.copy_params(pv))
        assert npts[b] == len(want)
        got = packed[off[b]: off[b + 1]]
        assert got.tobytes() == xyzi[b, : npts[b]].tobytes()
        if len(want):
            assert np.max(np.abs(got[:, :2].astype(np.float64) - want[:, :2])) <= XYZ_TOL
            assert got[:, 3].tobytes() == want[:, 3].tobytes()

    # ascend batch, in place
    d_nodes2 = d_nodes.clone()
    gpu.ascend_batch_dev(d_nodes2.data_ptr(), n, d_len.data_ptr(), B, d_st.data_ptr())
    gpu.synchronize()
    asc = d_nodes2.cpu().numpy().view(NODE_DTYPE).reshape(B, n)
    st = d_st.cpu().numpy()
    for b in range(B):
        want, res = oracle.ascend(batch[b, : lens[b]])
        assert (st[b] & abi.SCAN_ALL_INVALID != 0) == (res != 0)
        if res == 0:
            assert oracle_lib.canon_equal_angle_runs(asc[b, : lens[b]]).tobytes() == \
                oracle_lib.canon_equal_angle_runs(want).tobytes()
        assert asc[b, lens[b]:].tobytes() == batch[b, lens[b]:].tobytes()  # tail untouched


def test_scan_length_is_clamped_to_its_slot(gpu):
    """A length word larger than the slot (caller bug) must not make a kernel read the next
    scan: every batch kernel uses min(n_per_scan[b], n_stride)."""
    torch = _torch()
    B, n = 6, 2000
    batch = synth.make_batch(5, B, n, jitter=1)
    dev = torch.device("cuda:0")
    d_nodes = torch.from_numpy(batch.view(np.uint8).reshape(B, n * 8)).to(dev)
    outs = []
    for extra in (0, 777):
        d_len = torch.full((B,), n + extra, dtype=torch.int32, device=dev)
        d_r = torch.zeros(B, n, dtype=torch.float32, device=dev)
        d_i = torch.zeros(B, n, dtype=torch.float32, device=dev)
        d_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
        p = Params.defaults(range_max=40.0)
        gpu.laserscan_batch_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, p, d_r.data_ptr(),
                                d_i.data_ptr(), d_cnt.data_ptr())
        pv = Params.defaults(clip_enable=1, range_max=40.0, voxel_enable=1, ror_enable=1)
        d_xyzi = torch.zeros(B, n, 4, dtype=torch.float32, device=dev)
        d_np = torch.zeros(B, dtype=torch.int32, device=dev)
        d_st = torch.zeros(B, dtype=torch.int32, device=dev)
        gpu.cloud_batch_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, pv, d_xyzi.data_ptr(), n,
                            d_np.data_ptr(), d_st.data_ptr())
        d_asc = d_nodes.clone()
        gpu.ascend_batch_dev(d_asc.data_ptr(), n, d_len.data_ptr(), B, d_st.data_ptr())
        gpu.synchronize()
        outs.append([t.cpu().numpy().tobytes() for t in (d_r, d_i, d_cnt, d_xyzi, d_np, d_asc)])
    assert outs[0] == outs[1]


def test_c5_eight_sensors_fused_cloud(gpu, oracle):
    """BASELINE config 5 on one device: 8 sensors x one 32 000-sample scan each, E1 clip ->
    E5 radius-outlier removal -> E4 voxel grid, packed into ONE fused cloud (sensor order;
    the reference only ever publishes the identity base_link->frame_id transform,
    src/rplidar_node.cpp:183-197)."""
    torch = _torch()
    S, n = 8, 32000
    batch = np.stack([synth.make_scan(500 + s, 0, n, noise_m=0.03, jitter=2) for s in range(S)])
    dev = torch.device("cuda:0")
    d_nodes = torch.from_numpy(batch.view(np.uint8).reshape(S, n * 8)).to(dev)
    d_len = torch.full((S,), n, dtype=torch.int32, device=dev)
    p = Params.defaults(clip_enable=1, range_max=40.0, ror_enable=1, voxel_enable=1)
    out_stride = 16384
    d_xyzi = torch.zeros(S, out_stride, 4, dtype=torch.float32, device=dev)
    d_np = torch.zeros(S, dtype=torch.int32, device=dev)
    d_st = torch.zeros(S, dtype=torch.int32, device=dev)
    d_off = torch.zeros(S + 1, dtype=torch.int64, device=dev)
    d_packed = torch.zeros(S * out_stride, 4, dtype=torch.float32, device=dev)
    gpu.cloud_batch_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), S, p, d_xyzi.data_ptr(),
                        out_stride, d_np.data_ptr(), d_st.data_ptr())
    gpu.pack_clouds_dev(d_xyzi.data_ptr(), out_stride, d_np.data_ptr(), S, d_packed.data_ptr(),
                        d_off.data_ptr())
    gpu.synchronize()
    assert int(d_st.max()) == 0
    off = d_off.cpu().numpy()
    fused = d_packed.cpu().numpy()[: off[S]]
    want = [oracle.cloud_pipeline(batch[s], oracle_lib.copy_params(p))[0] for s in range(S)]
    assert [int(off[s + 1] - off[s]) for s in range(S)] == [len(w) for w in want]
    ref = np.concatenate(want)
    assert np.max(np.abs(fused[:, :2].astype(np.float64) - ref[:, :2])) <= XYZ_TOL
    assert fused[:, 3].tobytes() == ref[:, 3].tobytes()


def test_full_size_properties(gpu):
    """BASELINE config 3 shape (reduced batch so the oracle is not needed): size-independent
    properties — ascend output sorted + idempotent multiset, Mode A beam_count == #valid,
    every finite range is one of the scan's dist values, voxel count conservation."""
    torch = _torch()
    B, n = 256, 32000
    batch = synth.make_batch(33, B, n)
    dev = torch.device("cuda:0")
    d_nodes = torch.from_numpy(batch.view(np.uint8).reshape(B, n * 8)).to(dev)
    d_len = torch.full((B,), n, dtype=torch.int32, device=dev)
    d_st = torch.zeros(B, dtype=torch.int32, device=dev)
    gpu.ascend_batch_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, d_st.data_ptr())
    d_r = torch.empty(B, n, dtype=torch.float32, device=dev)
    d_i = torch.empty(B, n, dtype=torch.float32, device=dev)
    d_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    p = Params.defaults(range_max=40.0)
    gpu.laserscan_batch_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, p, d_r.data_ptr(),
                            d_i.data_ptr(), d_cnt.data_ptr())
    gpu.synchronize()
    asc = d_nodes.cpu().numpy().view(NODE_DTYPE).reshape(B, n)
    assert np.all(np.diff(asc["angle_z_q14"].astype(np.int32), axis=1) >= 0)
    valid = batch["dist_mm_q2"] != 0
    assert np.array_equal(np.sort(asc["dist_mm_q2"], axis=1), np.sort(batch["dist_mm_q2"], axis=1))
    cnt = d_cnt.cpu().numpy()
    assert np.array_equal(cnt, valid.sum(1))
    r = d_r.cpu().numpy()
    for b in range(0, B, 37):
        rr = r[b, : cnt[b]]
        fin = rr[np.isfinite(rr)]
        dm = (batch[b]["dist_mm_q2"][valid[b]].astype(np.float32) / np.float32(4000.0))
        assert np.isin(fin, dm).all()
        assert len(fin) > 0.5 * cnt[b]
    # voxel conservation, all 256 scans, no C oracle: the plain cloud (E1 + E2, bit-exact by the
    # other tests) grouped by cell in numpy must give the voxel kernel's cells — same number, same
    # (iy, ix) order, centroids = fp64 means of the members, intensity = fp64 mean; every kept
    # sample is in exactly one cell (the counts add up)
    d_raw = torch.from_numpy(batch.view(np.uint8).reshape(B, n * 8)).to(dev)
    pc = Params.defaults(clip_enable=1, range_max=40.0)
    d_pts = torch.zeros(B, n, 4, dtype=torch.float32, device=dev)
    d_npc = torch.zeros(B, dtype=torch.int32, device=dev)
    d_stc = torch.zeros(B, dtype=torch.int32, device=dev)
    gpu.cloud_batch_dev(d_raw.data_ptr(), n, d_len.data_ptr(), B, pc, d_pts.data_ptr(), n,
                        d_npc.data_ptr(), d_stc.data_ptr())
    pv = Params.defaults(clip_enable=1, range_max=40.0, voxel_enable=1)
    d_vox = torch.zeros(B, 8192, 4, dtype=torch.float32, device=dev)
    d_npv = torch.zeros(B, dtype=torch.int32, device=dev)
    d_stv = torch.zeros(B, dtype=torch.int32, device=dev)
    gpu.cloud_batch_dev(d_raw.data_ptr(), n, d_len.data_ptr(), B, pv, d_vox.data_ptr(), 8192,
                        d_npv.data_ptr(), d_stv.data_ptr())
    gpu.synchronize()
    assert int(d_stc.max()) == 0 and int(d_stv.max()) == 0
    pts, npc = d_pts.cpu().numpy(), d_npc.cpu().numpy()
    vox, npv = d_vox.cpu().numpy(), d_npv.cpu().numpy()
    dmf = batch["dist_mm_q2"].astype(np.float32) / np.float32(4000.0)
    kept = valid & (dmf >= np.float32(0.15)) & (dmf <= np.float32(40.0))
    assert np.array_equal(npc, kept.sum(1))
    leaf = np.float32(0.05)
    for b in range(B):
        pb = pts[b, : npc[b]]
        ix = np.floor(pb[:, 0] / leaf).astype(np.int64)
        iy = np.floor(pb[:, 1] / leaf).astype(np.int64)
        key = iy * 65536 + ix
        order = np.argsort(key, kind="stable")
        ks = key[order]
        starts = np.r_[0, np.flatnonzero(np.diff(ks)) + 1]
        counts = np.diff(np.r_[starts, len(ks)])
        assert npv[b] == len(starts) and counts.sum() == npc[b]
        # the spec's centroid: fp64 sum in sample order / count, rounded to float32
        mean = lambda col: (np.add.reduceat(pb[order, col].astype(np.float64), starts)
                            / counts).astype(np.float32)
        vb = vox[b, : npv[b]]
        assert np.max(np.abs(vb[:, 0].astype(np.float64) - mean(0))) <= XYZ_TOL
        assert np.max(np.abs(vb[:, 1].astype(np.float64) - mean(1))) <= XYZ_TOL
        assert np.array_equal(vb[:, 3], mean(3))


# ------------------------------------------------------------ directly against the reference
def test_hip_path_matches_genuine_reference_golden(gpu):
    """No oracle in between: the HIP path against what the GENUINE reference code produced for
    the same inputs (tests/golden/*.npz, generated by tests/golden/make_golden.py from
    /root/reference: the real SDK ascendScanData and the real RPlidarNode::publish_scan)."""
    from pathlib import Path
    from tests.cases import GOLDEN_CASES
    gold = Path(__file__).resolve().parent / "golden"
    ga = np.load(gold / "ascend_golden.npz")
    gp = np.load(gold / "publish_scan_golden.npz")
    checked = 0
    for name in GOLDEN_CASES:
        nodes = CASES[name]
        assert ga[f"{name}__in"].tobytes() == nodes.tobytes(), "case generator drifted"
        asc = nodes.copy()
        res = gpu.ascend(asc)
        assert res == int(ga[f"{name}__res"]), name
        if res == 0:  # equal-angle runs: upstream's order is introsort's, compare canonically
            assert oracle_lib.canon_equal_angle_runs(asc).tobytes() == \
                oracle_lib.canon_equal_angle_runs(ga[f"{name}__out"]).tobytes(), name
        v = nodes[nodes["dist_mm_q2"] != 0]
        unique = len(np.unique(v["angle_z_q14"])) == len(v)
        for kind in (0, 1, 2):  # Dummy / Real OLD_TYPE / Real NEW_TYPE
            for inv in (0, 1):
                for sp in (0, 1):
                    p = Params.defaults(is_new_protocol=int(kind == 2), inverted=inv,
                                        scan_processing=sp, range_max=40.0)
                    r, i, m = gpu.scan_to_laserscan(nodes, p, 0.125)
                    tag = f"{name}__k{kind}_i{inv}_s{sp}"
                    assert bytes(m) == gp[tag + "__meta"].tobytes(), tag
                    if not m.published:
                        continue
                    if sp or unique:  # Mode A ranges never depend on the tie order
                        assert r.tobytes() == gp[tag + "__ranges"].tobytes(), tag
                    if unique:
                        assert i.tobytes() == gp[tag + "__intens"].tobytes(), tag
                    checked += 1
    assert checked > 100
    gd = np.load(gold / "dummy_golden.npz")  # config 1: the real DummyLidarDriver's scans and
    for k in range(3):                        # what the real node publishes for them, byte for byte
        nodes = gd[f"scan{k}"].view(NODE_DTYPE).reshape(-1)
        for inv in (0, 1):
            for sp in (0, 1):
                p = Params.defaults(inverted=inv, scan_processing=sp, range_max=40.0)
                r, i, m = gpu.scan_to_laserscan(nodes, p, 0.1)
                tag = f"scan{k}__i{inv}_s{sp}"
                assert bytes(m) == gd[tag + "__meta"].tobytes(), tag
                assert m.count == 360
                assert r.tobytes() == gd[tag + "__ranges"].tobytes(), tag
                assert i.tobytes() == gd[tag + "__intens"].tobytes(), tag


@pytest.mark.parametrize("name", __import__("tests.cases", fromlist=["x"]).LARGE_GOLDEN_CASES)
def test_hip_path_matches_genuine_reference_large(gpu, name):
    """Config 2 (32 000 samples) and the other large cases directly against the genuine SDK /
    node outputs (SHA-256 digests in tests/golden/large_golden.npz; canonical forms where the
    reference's order inside equal-angle runs is introsort's)."""
    from pathlib import Path
    from tests import canon
    g = np.load(Path(__file__).resolve().parent / "golden" / "large_golden.npz")

    def asc(nodes):
        out = nodes.copy()
        return out, gpu.ascend(out)

    def ls(nodes, kind, inv, sp):
        p = Params.defaults(is_new_protocol=int(kind == 2), inverted=inv, scan_processing=sp,
                            range_max=40.0)
        r, i, m = gpu.scan_to_laserscan(nodes, p, 0.125)
        return r, i, bytes(m)

    assert canon.check_large_golden(g, name, CASES[name], asc, ls) == 12


@pytest.mark.parametrize("n", [700, 20000, 32000])
def test_laserscan_far_distances(gpu, oracle, n):
    """Distances far beyond anything a lidar reports: above 2^24 two dist_mm_q2 values may share a
    float, and the reference compares the floats (the winner of a bin is the first of equal
    dist_m in angle order).  Just below 2^22, around it, around 2^24, at the top of the u32 range,
    and a scan where every kept sample is that far — against the oracle, bit for bit; then the
    same scans through the batch entry point, more of them than compute units, next to ordinary
    and empty ones in every order of succession."""
    rng = np.random.default_rng(900 + n)
    base = synth.make_scan(31, n, n, invalid_p=0.05, kind="ring", r0_range=(2.0, 20.0))
    variants = {}
    near = base.copy()  # just below the seam
    near["dist_mm_q2"][rng.integers(0, n, 40)] = (1 << 22) - 1 - rng.integers(0, 3, 40)
    variants["below"] = near
    far = base.copy()   # a few samples at / beyond the seam, some sharing a float (spacing 1 above 2^24)
    k = rng.integers(0, n, 64)
    far["dist_mm_q2"][k[:16]] = (1 << 22) + rng.integers(0, 4, 16)
    far["dist_mm_q2"][k[16:40]] = (1 << 24) + rng.integers(0, 6, 24)
    far["dist_mm_q2"][k[40:]] = np.uint32(0xFFFFFFF0) + rng.integers(0, 15, 24).astype(np.uint32)
    variants["beyond"] = far
    allfar = base.copy()  # every kept sample far, many equal floats in a bin
    nz = allfar["dist_mm_q2"] != 0
    allfar["dist_mm_q2"][nz] = (1 << 25) + (rng.integers(0, 8, int(nz.sum())) & ~np.uint32(1))
    variants["all_far"] = allfar
    for name, nodes in variants.items():
        for inverted in (0, 1):
            p = Params.defaults(is_new_protocol=1, inverted=inverted, scan_processing=1,
                                clip_enable=0, range_max=40.0)
            wr, wi, wm = oracle.publish_scan(nodes, oracle_lib.copy_params(p), 0.1)
            gr, gi, gm = gpu.scan_to_laserscan(nodes, p, 0.1)
            assert bytes(gm) == bytes(wm), name
            assert gr.tobytes() == wr.tobytes(), name
            if not _has_intensity_tie(nodes, p):
                assert gi.tobytes() == wi.tobytes(), name
    if n != 700:
        return
    # batch: more scans than compute units — ordinary, far, empty, all far, just below
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    kinds = [base, variants["beyond"], np.zeros(n, NODE_DTYPE), variants["all_far"], variants["below"]]
    B = 5 * 256 + 7
    which = [(i // 256 + 3 * (i % 256)) % 5 for i in range(B)]
    p = Params.defaults(is_new_protocol=1, scan_processing=1, clip_enable=0, range_max=40.0)
    h = np.zeros((B, n), NODE_DTYPE)
    for i, kd in enumerate(which):
        h[i] = kinds[kd]
    d_nodes = torch.from_numpy(h.view(np.uint8).reshape(B, n * 8)).to(dev)
    d_len = torch.full((B,), n, dtype=torch.int32, device=dev)
    d_r = torch.zeros(B, n, dtype=torch.float32, device=dev)
    d_i = torch.zeros(B, n, dtype=torch.float32, device=dev)
    d_c = torch.zeros(B, dtype=torch.int32, device=dev)
    gpu.laserscan_batch_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, p, d_r.data_ptr(),
                            d_i.data_ptr(), d_c.data_ptr())
    gpu.synchronize()
    want = [oracle.publish_scan(kd, oracle_lib.copy_params(p), 0.1) for kd in kinds]
    cnt, rr = d_c.cpu().numpy(), d_r.cpu().numpy()
    for i, kd in enumerate(which):
        wr, _, wm = want[kd]
        assert int(cnt[i]) == wm.count, i
        assert rr[i, :wm.count].tobytes() == wr[:wm.count].tobytes(), i


def test_mode_a_tie_rule(gpu, oracle):
    """The reference keeps the first minimum in std::sort order (:657), i.e. introsort decides
    among samples of one bin with identical dist_m.  The device rule is stated, deterministic
    and tested here: the winner of a bin is the minimum over (dist_m, angle word, intensity)."""
    rng = np.random.default_rng(77)
    n = 4096
    nodes = np.zeros(n, NODE_DTYPE)
    nodes["angle_z_q14"] = np.sort(rng.integers(0, 65536, n) & ~np.uint16(7))  # many equal words
    nodes["dist_mm_q2"] = rng.choice([4000, 4000, 4001, 8000, 0], n)  # 4000/4001: same float? no
    nodes["quality"] = rng.integers(0, 256, n)
    for is_new in (0, 1):
        for inv in (0, 1):
            p = Params.defaults(is_new_protocol=is_new, inverted=inv, range_max=40.0)
            gr, gi, gm = gpu.scan_to_laserscan(nodes, p, 0.1)
            wr, _, wm = oracle.publish_scan(nodes, oracle_lib.copy_params(p), 0.1)
            assert bytes(gm) == bytes(wm) and gr.tobytes() == wr.tobytes()
            v = nodes[nodes["dist_mm_q2"] != 0]
            cnt = len(v)
            q = v["angle_z_q14"].astype(np.float32)
            ang = (q * np.float32(90.0) / np.float32(16384.0)).astype(np.float64)
            ang = (ang * (np.pi / np.float64(np.float32(180.0)))).astype(np.float32)  # :588-589
            if inv:  # :646-651
                ang = (np.float64(np.float32(2.0)) * np.pi - ang.astype(np.float64)).astype(np.float32)
                ang = np.where(ang.astype(np.float64) >= 2.0 * np.pi,
                               (ang.astype(np.float64) - 2.0 * np.pi).astype(np.float32), ang)
            inc = np.float32((2.0 * np.pi) / np.float64(cnt))  # :635
            idx = (ang / inc).astype(np.int32)  # :653-654, float32 divide, truncation
            dm = (v["dist_mm_q2"].astype(np.float32) / np.float32(4000.0))
            inten = (v["quality"] if is_new else (v["quality"] >> 2)).astype(np.int64)
            key = (dm.view(np.uint32).astype(np.int64) << 24) | \
                  (v["angle_z_q14"].astype(np.int64) << 8) | inten
            ok = (idx >= 0) & (idx < cnt)
            want_i = np.zeros(cnt, np.float32)
            best = np.full(cnt, np.iinfo(np.int64).max)
            np.minimum.at(best, idx[ok], key[ok])
            hit = best != np.iinfo(np.int64).max
            want_i[hit] = (best[hit] & 0xFF).astype(np.float32)
            assert np.array_equal(np.isfinite(gr), hit)
            assert gi.tobytes() == want_i.tobytes()


# ----------------------------------------------------- a-ext against the SECOND writer's vectors
def _ext_gold():
    from tests.golden.make_ext_golden import EXT_PARAM_SETS, FULL_MAX_POINTS
    gold = np.load(oracle_lib.ROOT / "tests" / "golden" / "ext_golden.npz")
    return gold, EXT_PARAM_SETS, FULL_MAX_POINTS


@pytest.mark.parametrize("tag", ["cloud", "cloud_inv_new", "cloud_q48", "voxel", "voxel_inv",
                                 "voxel_leaf10_q48", "ror_voxel", "ror_cloud"])
def test_hip_path_matches_second_writer_golden(gpu, tag):
    """The extension kernels (E1 / E2 / E4 / E5) against tests/golden/ext_golden.npz: the committed
    outputs of oracle/ext_second_writer.py, an independent numpy / scipy writer of SURVEY.md
    §8(a-ext) — vectors that neither the kernels' author's C++ oracle nor the kernels produced.
    Small cases are compared point by point (count, intensity bits, x / y within 1e-6 m; the
    unvoxelised cloud bit for bit), every case by its point count."""
    gold, sets, full_max = _ext_gold()
    kw = dict(sets)[tag]
    p = Params.defaults(**{k: (int(v) if isinstance(v, bool) else v) for k, v in kw.items()})
    checked = 0
    for name, nodes in CASES.items():
        key = f"{name}__{tag}"
        if key + "__n" not in gold:
            continue  # (ROR on the largest cases is not in the file)
        got, status = gpu.scan_to_cloud(nodes, p)
        assert status == 0, key
        assert len(got) == int(gold[key + "__n"]), key
        if key + "__pts" not in gold:
            continue
        want = gold[key + "__pts"]
        assert got[:, 3].tobytes() == want[:, 3].tobytes(), key
        assert np.all(got[:, 2] == 0), key
        if len(want):
            assert np.max(np.abs(got[:, :2].astype(np.float64) - want[:, :2])) <= XYZ_TOL, key
        if not kw.get("voxel_enable"):
            assert got.tobytes() == want.tobytes(), key
        checked += 1
    assert checked >= 20


# ------------------------------------------------- ascendScanData on almost-ascending scans
@pytest.mark.parametrize("jitter", [1, 2, 3, 5, 8, 12, 40])
def test_ascend_batch_local_repair_regimes(gpu, oracle, jitter):
    """What a real sensor delivers: measured angles jitter and the interpolated angles of invalid nodes
    (src/sdk/src/sl_lidar_driver.cpp:171-178) do not mesh with their neighbours, so the filled scan is
    almost, not quite, ascending.  Small disorder is repaired inside the streaming kernel (odd-even
    transposition in registers, chunk boundaries in 16-sample windows), larger disorder falls back
    to the sorting kernel; either way the result must be ascendScanData's, with this library's tie
    rule (equal angle words keep their input order)."""
    torch = _torch()
    B, n = 40, 32000
    batch = synth.make_batch(4242 + jitter, B, n, jitter=jitter)
    lens = np.full(B, n, np.uint32)
    lens[1], lens[2], lens[3], lens[4] = 31999, 129, 128, 127  # odd length, chunk-sized scans
    lens[5] = 8191
    dev = torch.device("cuda:0")
    d_nodes = torch.from_numpy(batch.view(np.uint8).reshape(B, n * 8).copy()).to(dev)
    d_len = torch.from_numpy(lens.astype(np.int32)).to(dev)
    d_st = torch.zeros(B, dtype=torch.int32, device=dev)
    gpu.ascend_batch_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, d_st.data_ptr())
    gpu.synchronize()
    asc = d_nodes.cpu().numpy().view(NODE_DTYPE).reshape(B, n)
    assert np.all(d_st.cpu().numpy() == 0)
    for b in range(B):
        src = batch[b, : lens[b]]
        want, res = oracle.ascend(src)
        got = asc[b, : lens[b]]
        assert res == 0
        assert np.array_equal(got["angle_z_q14"], want["angle_z_q14"]), (jitter, b)
        assert oracle_lib.canon_equal_angle_runs(got).tobytes() == \
            oracle_lib.canon_equal_angle_runs(want).tobytes(), (jitter, b)
        # stability: the valid samples (whose angle words the fill never touches) in stable order
        v = src[src["dist_mm_q2"] != 0]
        assert got[got["dist_mm_q2"] != 0].tobytes() == \
            v[np.argsort(v["angle_z_q14"], kind="stable")].tobytes(), (jitter, b)
        assert asc[b, lens[b]:].tobytes() == batch[b, lens[b]:].tobytes()  # the slot's tail untouched


def test_ascend_then_laserscan_equals_laserscan(gpu, oracle):
    """rplgpu_ascend_laserscan_batch_dev: grab_scan_data (ascendScanData in place,
    src/lidar_driver_wrapper.cpp:328-337) followed by publish_scan (src/rplidar_node.cpp:568-680) in
    one pass.  publish_scan drops the invalid nodes — the only ones ascendScanData rewrites — and
    sorts by angle itself, so the LaserScan must be the one of the raw nodes, bit for bit, in every
    mode; checked against the two-step pipeline on the device and against the oracle's two steps."""
    torch = _torch()
    B, n = 32, 8000
    dev = torch.device("cuda:0")
    for jitter, kw in ((0, {}), (3, {}), (50, {"rotate": True}), (0, {"kind": "uniform"})):
        batch = synth.make_batch(99 + jitter, B, n, jitter=jitter, **kw)
        batch[5]["dist_mm_q2"] = 0  # an all-invalid scan: ascend fails, nothing is published
        lens = np.array([n - 7 * b for b in range(B)], np.uint32)
        d_len = torch.from_numpy(lens.astype(np.int32)).to(dev)
        for sp in (1, 0):
            for inv in (0, 1):
                p = Params.defaults(range_max=40.0, scan_processing=sp, inverted=inv)
                d_raw = torch.from_numpy(batch.view(np.uint8).reshape(B, n * 8).copy()).to(dev)
                out = [[torch.zeros(B, n, dtype=torch.float32, device=dev) for _ in range(2)] +
                       [torch.zeros(B, dtype=torch.int32, device=dev)] for _ in range(3)]
                d_st = torch.full((B,), -1, dtype=torch.int32, device=dev)
                # (a) one pass, nodes only read
                before = d_raw.clone()
                gpu.ascend_laserscan_batch_dev(d_raw.data_ptr(), n, d_len.data_ptr(), B, p,
                                               out[0][0].data_ptr(), out[0][1].data_ptr(),
                                               out[0][2].data_ptr())
                gpu.synchronize()
                assert torch.equal(before, d_raw) and int(d_st.max()) == -1
                # (b) two steps: ascend in place, then the LaserScan of the ascended nodes
                d_two = d_raw.clone()
                gpu.ascend_batch_dev(d_two.data_ptr(), n, d_len.data_ptr(), B, 0)
                gpu.laserscan_batch_dev(d_two.data_ptr(), n, d_len.data_ptr(), B, p,
                                        out[1][0].data_ptr(), out[1][1].data_ptr(), out[1][2].data_ptr())
                # (c) one call that also leaves the ascended nodes
                d_both = d_raw.clone()
                gpu.ascend_laserscan_batch_dev(d_both.data_ptr(), n, d_len.data_ptr(), B, p,
                                               out[2][0].data_ptr(), out[2][1].data_ptr(),
                                               out[2][2].data_ptr(), True, d_st.data_ptr())
                gpu.synchronize()
                assert torch.equal(d_both, d_two)
                st = d_st.cpu().numpy()
                assert st[5] & abi.SCAN_ALL_INVALID and np.all(np.delete(st, 5) == 0)
                cnt = out[0][2].cpu().numpy()
                assert cnt[5] == 0
                for o in out[1:]:
                    assert np.array_equal(o[2].cpu().numpy(), cnt)
                r0, i0 = out[0][0].cpu().numpy(), out[0][1].cpu().numpy()
                for o in out[1:]:
                    r, i = o[0].cpu().numpy(), o[1].cpu().numpy()
                    for b in range(B):
                        assert r[b, : cnt[b]].tobytes() == r0[b, : cnt[b]].tobytes(), (jitter, sp, inv, b)
                        assert i[b, : cnt[b]].tobytes() == i0[b, : cnt[b]].tobytes(), (jitter, sp, inv, b)
                # the oracle's two steps on a few scans (ties canonicalised as everywhere: Mode A
                # ranges are order-free, the rest is compared where no angle word repeats)
                for b in (0, 7, 31):
                    src = batch[b, : lens[b]]
                    asc_nodes, res = oracle.ascend(src)
                    wr, wi, wm = oracle.publish_scan(asc_nodes if res == 0 else src,
                                                     oracle_lib.copy_params(p), 0.1)
                    assert cnt[b] == wm.count
                    if sp == 1:
                        assert r0[b, : cnt[b]].tobytes() == wr.tobytes()


def test_ascend_wrap_zone(gpu, oracle):
    """Invalid nodes at the end of a scan whose interpolated angle passes 360 degrees are given an
    angle near zero (src/sdk/src/sl_lidar_driver.cpp:174-176) and belong at the FRONT of the sorted
    scan: every other node moves up.  The streaming kernel does that move itself when the wrap
    zone (at most 64 indices) lies inside the scan's last 128-sample chunk, whatever the pattern of
    valid and invalid nodes in it; other shapes go to the sorting kernel.  Both must give
    ascendScanData's result."""
    torch = _torch()
    n = 32000
    # (angle offset of the whole scan / of sample 0 alone, indices made invalid, indices forced valid, length)
    shapes = []
    for q0, k in ((5, 1), (9, 2), (40, 5), (200, 64), (200, 65), (131, 40)):
        shapes.append((q0, 0, list(range(n - k, n)), [], n))       # the wrapped fills go to the very front
    for q0, k in ((5, 1), (40, 5), (200, 64)):
        shapes.append((0, q0, list(range(n - k, n)), [], n))       # ... or in between the first samples
    shapes.append((40, 0, [n - 1, n - 2, n - 4], [n - 3], n))      # a valid node inside the zone
    shapes.append((60, 0, [n - 30, n - 17, n - 16, n - 3], [n - 1, n - 2, n - 4, n - 5], n))  # scattered fills
    shapes.append((120, 0, list(range(n - 58, n, 2)), list(range(n - 57, n, 2)), n))          # every other node
    shapes.append((3000, 0, list(range(n - 30, n)), [], n))        # a zone of ~1500 samples
    shapes.append((40, 0, list(range(20000 - 3, 20000)), [], 20000))  # shorter scans
    shapes.append((40, 0, [298, 299], [], 300))
    shapes.append((40, 0, [128], [], 129))
    shapes.append((40, 0, [127], [], 128))
    shapes.append((0, 0, list(range(n - 4, n)), [], n))            # front 0: nothing wraps
    for jitter in (0, 3):
        B = len(shapes)
        batch = synth.make_batch(777 + jitter, B, n, jitter=jitter)
        lens = np.zeros(B, np.uint32)
        for b, (qall, q0, inv, val, ln) in enumerate(shapes):
            lens[b] = ln
            batch[b]["angle_z_q14"] = np.minimum(batch[b]["angle_z_q14"].astype(np.uint32) + qall, 65535)
            if q0:
                batch[b]["angle_z_q14"][0] = q0
            if batch[b]["dist_mm_q2"][0] == 0:
                batch[b]["dist_mm_q2"][0] = 4000
            batch[b]["dist_mm_q2"][inv] = 0
            for v in val:
                batch[b]["dist_mm_q2"][v] = 8000
        dev = torch.device("cuda:0")
        d_nodes = torch.from_numpy(batch.view(np.uint8).reshape(B, n * 8).copy()).to(dev)
        d_len = torch.from_numpy(lens.astype(np.int32)).to(dev)
        d_st = torch.zeros(B, dtype=torch.int32, device=dev)
        gpu.ascend_batch_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, d_st.data_ptr())
        gpu.synchronize()
        asc = d_nodes.cpu().numpy().view(NODE_DTYPE).reshape(B, n)
        assert np.all(d_st.cpu().numpy() == 0)
        moved_front = 0
        for b in range(B):
            src = batch[b, : lens[b]]
            want, res = oracle.ascend(src)
            got = asc[b, : lens[b]]
            assert res == 0
            assert np.array_equal(got["angle_z_q14"], want["angle_z_q14"]), (jitter, b)
            assert oracle_lib.canon_equal_angle_runs(got).tobytes() == \
                oracle_lib.canon_equal_angle_runs(want).tobytes(), (jitter, b)
            v = src[src["dist_mm_q2"] != 0]
            assert got[got["dist_mm_q2"] != 0].tobytes() == \
                v[np.argsort(v["angle_z_q14"], kind="stable")].tobytes(), (jitter, b)
            assert asc[b, lens[b]:].tobytes() == batch[b, lens[b]:].tobytes()
            moved_front += int(got["dist_mm_q2"][0] == 0 and src["dist_mm_q2"][0] != 0)
        assert moved_front >= 5  # the cases really exercise the wrap


def _filled_words(src):
    """The angle word ascendScanData gives every node BEFORE it sorts (src/sdk/src/sl_lidar_driver.cpp:
    171-178), for scans whose node 0 is valid (no head chain): float32 operations one by one, as the SDK."""
    n = len(src)
    f32 = np.float32
    assert src["dist_mm_q2"][0] != 0
    inc = f32(360.0) / f32(n)
    front = f32(src["angle_z_q14"][0]) * f32(90.0) / f32(16384.0)
    i = np.arange(n, dtype=np.float32)
    e = front + i * inc
    e = np.where(e > f32(360.0), e - f32(360.0), e).astype(np.float32)
    q = ((e * f32(16384.0) / f32(90.0)).astype(np.uint32) & 0xFFFF).astype(np.uint16)
    return np.where(src["dist_mm_q2"] == 0, q, src["angle_z_q14"])


def test_ascend_tie_rule_with_wrapped_fills(gpu, oracle):
    """This library's tie rule — nodes of equal angle word keep their INPUT order — where it is hardest
    to keep: wrapped fills (the last nodes of the input, moved to the front of the result)
    against nodes of equal word and smaller index, in the shapes that end in the sorting kernel: sixteen
    or more wrapped fills whose last word equals the first node's, and a few wrapped fills in a scan
    that is out of order somewhere else.  Expected = the input with its filled words, sorted stably."""
    torch = _torch()
    n = 32000
    cases = []
    # (k wrapped fills; the scan's angle offset q0 keeps the wrap zone — q0 / 2.05 indices — within the 64 the
    # streaming kernel handles itself; far: disorder elsewhere)
    for k, q0, far in ((16, 40, False), (40, 100, False), (60, 128, False), (3, 20, True), (12, 40, True),
                       (20, 50, True), (1, 10, True)):
        scan = synth.make_scan(4321 + k, 0, n, invalid_p=0.05)
        scan["angle_z_q14"] = np.minimum(scan["angle_z_q14"].astype(np.uint32) + q0, 65535).astype(np.uint16)
        scan["dist_mm_q2"][0] = 4000
        scan["dist_mm_q2"][n - k - 8: n - k] = 5000  # (valid nodes in front of the zone)
        scan["dist_mm_q2"][n - k:] = 0
        w = _filled_words(scan)
        # a valid node at the scan's front gets the word of the LAST wrapped fill: a tie with the smaller index
        # (node 0 itself cannot: the fills are interpolated from its angle and always come out below it)
        scan["dist_mm_q2"][1] = 4400
        scan["angle_z_q14"][1] = w[n - 1]
        assert np.array_equal(_filled_words(scan)[n - k:], w[n - k:]) and w[n - 1] < q0  # it did wrap
        w = _filled_words(scan)
        zone = np.flatnonzero((w < q0) & (np.arange(n) > n // 2))  # indices whose interpolated angle wrapped
        assert zone.min() >= n - 64 and np.all(scan["dist_mm_q2"][zone.min(): n - k] != 0)  # the fills ARE the zone's invalid nodes
        if far:  # disorder no local repair reaches: the scan goes to the sorting kernel whatever the front does
            a, b = 9000, 9400
            scan["dist_mm_q2"][[a, b]] = 6000
            scan["angle_z_q14"][[a, b]] = scan["angle_z_q14"][[b, a]]
            # ... and one of the nodes in there carries a wrapped fill's word: the same tie, far from the front
            scan["angle_z_q14"][a + 5] = w[n - 1 - (k // 2)]
            scan["dist_mm_q2"][a + 5] = 7000
        cases.append(scan)
    cases = cases + cases  # (more than eight scans: batches of up to eight long scans take k_ascend<false>,
    B = len(cases)         #  not the streaming kernel this test is about)
    assert B > 8
    batch = np.stack(cases)
    dev = torch.device("cuda:0")
    d_nodes = torch.from_numpy(batch.view(np.uint8).reshape(B, n * 8).copy()).to(dev)
    d_len = torch.full((B,), n, dtype=torch.int32, device=dev)
    d_st = torch.zeros(B, dtype=torch.int32, device=dev)
    gpu.ascend_batch_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, d_st.data_ptr())
    gpu.synchronize()
    asc = d_nodes.cpu().numpy().view(NODE_DTYPE).reshape(B, n)
    assert np.all(d_st.cpu().numpy() == 0)
    for b in range(B):
        src = batch[b]
        want, res = oracle.ascend(src)
        assert res == 0
        filled = src.copy()
        filled["angle_z_q14"] = _filled_words(src)
        assert np.array_equal(np.sort(filled["angle_z_q14"]), want["angle_z_q14"]), b  # (the emulation is the SDK's)
        expect = filled[np.argsort(filled["angle_z_q14"], kind="stable")]
        assert asc[b].tobytes() == expect.tobytes(), b
        assert oracle_lib.canon_equal_angle_runs(asc[b]).tobytes() == \
            oracle_lib.canon_equal_angle_runs(want).tobytes(), b
    # the single-scan seam takes the same kernels
    one = batch[1].copy()
    assert gpu.ascend(one) == 0
    f = batch[1].copy()
    f["angle_z_q14"] = _filled_words(batch[1])
    assert one.tobytes() == f[np.argsort(f["angle_z_q14"], kind="stable")].tobytes()


# ------------------------------------------------------------- the plain cloud of a batch (E1 + E2)
def test_plain_cloud_batch(gpu, oracle):
    """The unvoxelised cloud of a batch (k_cloud) against the oracle, bit for bit: ragged lengths
    (0, 1, odd, chunk-sized, 32 000), inversion, the quality / range clip, the E5 mask in front
    (ror_enable), the E6 de-skew, a region too small for the cloud (truncated and flagged), and a
    sub-batch giving the same bytes as the whole one."""
    import sys
    sys.path.insert(0, str(oracle_lib.ROOT / "oracle"))
    import fusion_oracle as fo
    torch = _torch()
    B, n = 24, 32000
    batch = synth.make_batch(515, B, n, jitter=2)
    batch[7] = synth.make_scan(515, 7, n, kind="uniform")
    lens = np.array([n, 0, 1, 2, 127, 128, 129, 2047, 2048, 2049, 4097, 31999] + [n - 311 * b for b in range(12)],
                    np.uint32)
    dev = torch.device("cuda:0")
    d_nodes = torch.from_numpy(batch.view(np.uint8).reshape(B, n * 8).copy()).to(dev)
    d_len = torch.from_numpy(lens.astype(np.int32)).to(dev)
    d_np = torch.zeros(B, dtype=torch.int32, device=dev)
    d_st = torch.zeros(B, dtype=torch.int32, device=dev)
    for kw in (dict(), dict(inverted=1, is_new_protocol=1), dict(clip_enable=1, q_min=40, range_min=0.5, range_max=20.0),
               dict(clip_enable=1, range_max=40.0, ror_enable=1, ror_radius=0.05, ror_min_neighbors=2)):
        if kw.get("ror_enable"):  # (the oracle's ROR is quadratic: short scans only)
            sub = np.minimum(lens, 3000).astype(np.uint32)
            sub[sub == 2049] = 2049
        else:
            sub = lens
        d_sub = torch.from_numpy(sub.astype(np.int32)).to(dev)
        p = Params.defaults(**kw)
        d_xyzi = torch.full((B, n, 4), -1.0, dtype=torch.float32, device=dev)
        gpu.cloud_batch_dev(d_nodes.data_ptr(), n, d_sub.data_ptr(), B, p, d_xyzi.data_ptr(), n,
                            d_np.data_ptr(), d_st.data_ptr())
        gpu.synchronize()
        got, npts = d_xyzi.cpu().numpy(), d_np.cpu().numpy()
        assert np.all(d_st.cpu().numpy() == 0)
        for b in range(B):
            want = oracle.scan_to_cloud(batch[b, : sub[b]], oracle_lib.copy_params(p))
            assert npts[b] == len(want), (kw, b)
            assert got[b, : npts[b]].tobytes() == want.tobytes(), (kw, b)
            assert np.all(got[b, npts[b]:] == -1.0), (kw, b)  # nothing written past the cloud
        # a sub-batch gives the same bytes
        d_x8 = torch.full((8, n, 4), -1.0, dtype=torch.float32, device=dev)
        gpu.cloud_batch_dev(d_nodes.data_ptr() + 8 * n * 8, n, d_sub.data_ptr() + 8 * 4, 8, p,
                            d_x8.data_ptr(), n, d_np.data_ptr(), d_st.data_ptr())
        gpu.synchronize()
        assert d_x8.cpu().numpy().tobytes() == got[8:16].tobytes(), kw
    # a region that is too small: cut at out_stride, flagged, nothing beyond it written
    p = Params.defaults()
    d_small = torch.full((B, 1000, 4), -1.0, dtype=torch.float32, device=dev)
    gpu.cloud_batch_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, p, d_small.data_ptr(), 1000,
                        d_np.data_ptr(), d_st.data_ptr())
    gpu.synchronize()
    sm, npts, st = d_small.cpu().numpy(), d_np.cpu().numpy(), d_st.cpu().numpy()
    for b in range(B):
        want = oracle.scan_to_cloud(batch[b, : lens[b]], oracle_lib.copy_params(p))
        assert npts[b] == min(len(want), 1000)
        assert (st[b] & abi.SCAN_OUT_TRUNCATED != 0) == (len(want) > 1000)
        assert sm[b, : npts[b]].tobytes() == want[: npts[b]].tobytes()
    # E6 de-skew in the streaming kernel
    motion = np.tile(np.array([[1.5, -0.7, 0.9, 2.0e-5]], np.float32), (B, 1))
    motion[0] = 0
    d_motion = torch.from_numpy(motion).to(dev)
    d_xyzi = torch.zeros(B, n, 4, dtype=torch.float32, device=dev)
    gpu.cloud_deskew_batch_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, p, d_motion.data_ptr(),
                               d_xyzi.data_ptr(), n, d_np.data_ptr(), d_st.data_ptr())
    gpu.synchronize()
    got, npts = d_xyzi.cpu().numpy(), d_np.cpu().numpy()
    for b in (0, 3, 7, 11, 23):
        nodes = batch[b, : lens[b]]
        plain = oracle.scan_to_cloud(nodes, oracle_lib.copy_params(p))
        want = fo.deskew_cloud(plain, np.flatnonzero(nodes["dist_mm_q2"] != 0), motion[b])
        assert npts[b] == len(want) and got[b, : npts[b]].tobytes() == want.tobytes(), b
