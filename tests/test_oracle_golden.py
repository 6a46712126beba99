"""CPU tests: pin the oracle (oracle/liboracle.so) against
  (1) the golden vectors in tests/golden/, which were produced by the GENUINE reference code
      (real SDK ascendScanData, real RPlidarNode::publish_scan, real DummyLidarDriver) with
      tests/golden/make_golden.py, and the two KATs of SURVEY.md §8(c);
  (2) the genuine reference libraries themselves when oracle/_ref/ is built (build
      container only; the GPU box has no /root/reference and skips those).
Everything here must match BIT FOR BIT, including the order std::sort leaves equal angles
in (same libstdc++ introsort, same comparator sequence)."""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest

from tests import oracle_lib
from tests import canon
from tests.cases import CASES, GOLDEN_CASES, LARGE_GOLDEN_CASES

GOLD = Path(__file__).resolve().parent / "golden"


def test_node_layout():
    assert oracle_lib.NODE.itemsize == 8
    assert oracle_lib.NODE.fields["dist_mm_q2"][1] == 2  # unaligned u32 at byte offset 2


def test_kat1_ascend(oracle):
    g = np.load(GOLD / "ascend_golden.npz")
    out, res = oracle.ascend(g["kat1__in"])
    assert res == 0 == int(g["kat1__res"])
    want = [(0, 0, 12), (0, 32000, 28), (16384, 24000, 20), (24576, 20000, 16),
            (24576, 0, 24), (40960, 0, 0), (40960, 12000, 8), (49152, 8000, 4)]
    got = [(int(a), int(d), int(q)) for a, d, q, _ in out.tolist()]
    assert got == want
    assert out.tobytes() == g["kat1__out"].tobytes()


def test_kat2_publish_scan(oracle):
    nodes = CASES["kat2"]
    r, i, m = oracle.publish_scan(nodes, oracle_lib.params(inverted=0), 0.1)
    assert m.count == 3 and r.tolist() == [1.0, np.float32(0.001), 2.0]
    r, i, m = oracle.publish_scan(nodes, oracle_lib.params(inverted=1), 0.1)
    assert r[0] == 1.0 and r[1] == np.float32(0.001) and np.isinf(r[2])


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_ascend_golden(oracle, name):
    g = np.load(GOLD / "ascend_golden.npz")
    assert g[f"{name}__in"].tobytes() == CASES[name].tobytes(), "case generator drifted"
    out, res = oracle.ascend(CASES[name])
    assert res == int(g[f"{name}__res"])
    assert out.tobytes() == g[f"{name}__out"].tobytes()


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_publish_scan_golden(oracle, name):
    g = np.load(GOLD / "publish_scan_golden.npz")
    nodes = CASES[name]
    assert g[f"{name}__in"].tobytes() == nodes.tobytes(), "case generator drifted"
    for kind in (0, 1, 2):
        for inv in (0, 1):
            for sp in (0, 1):
                p = oracle_lib.params(is_new_protocol=int(kind == 2), inverted=inv,
                                      scan_processing=sp, range_max=40.0)
                r, i, m = oracle.publish_scan(nodes, p, 0.125)
                tag = f"{name}__k{kind}_i{inv}_s{sp}"
                assert bytes(m) == g[tag + "__meta"].tobytes(), tag
                assert r.tobytes() == g[tag + "__ranges"].tobytes(), tag
                assert i.tobytes() == g[tag + "__intens"].tobytes(), tag


def test_dummy_generator_golden(oracle):
    g = np.load(GOLD / "dummy_golden.npz")
    for k in range(3):
        assert oracle.gen_dummy(k).tobytes() == g[f"scan{k}"].tobytes()


def test_dummy_publish_golden(oracle):
    """Config 1 end to end: the genuine node's LaserScan for the genuine Dummy scans."""
    g = np.load(GOLD / "dummy_golden.npz")
    for k in range(3):
        nodes = g[f"scan{k}"]
        for inv in (0, 1):
            for sp in (0, 1):
                p = oracle_lib.params(inverted=inv, scan_processing=sp, range_max=40.0)
                r, i, m = oracle.publish_scan(nodes, p, 0.1)
                tag = f"scan{k}__i{inv}_s{sp}"
                assert bytes(m) == g[tag + "__meta"].tobytes(), tag
                assert r.tobytes() == g[tag + "__ranges"].tobytes(), tag
                assert i.tobytes() == g[tag + "__intens"].tobytes(), tag


@pytest.mark.parametrize("name", LARGE_GOLDEN_CASES)
def test_large_golden(oracle, name):
    """8192 ... 32768-sample scans (config 2's 32 000 among them) against digests of the genuine
    SDK's / the genuine node's outputs."""
    g = np.load(GOLD / "large_golden.npz")

    def ls(nodes, kind, inv, sp):
        p = oracle_lib.params(is_new_protocol=int(kind == 2), inverted=inv, scan_processing=sp,
                              range_max=40.0)
        r, i, m = oracle.publish_scan(nodes, p, 0.125)
        return r, i, bytes(m)

    assert canon.check_large_golden(g, name, CASES[name], oracle.ascend, ls) == 12


def test_effective_max_range(oracle):
    f = oracle.lib.orc_effective_max_range
    assert f(0.0, 12.0) == 12.0 and f(8.0, 12.0) == 8.0 and f(50.0, 40.0) == 40.0


# ---- live cross-checks against the genuine reference (build container only) ------------
def test_live_ascend_vs_real_sdk(oracle, reflibs):
    if reflibs is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this box)")
    rng = np.random.default_rng(5)
    for t in range(200):
        n = int(rng.integers(1, 3000))
        x = np.zeros(n, oracle_lib.NODE)
        x["angle_z_q14"] = rng.integers(0, 65536, n)
        x["dist_mm_q2"] = rng.integers(0, 200000, n) * (rng.random(n) > rng.random())
        x["quality"] = rng.integers(0, 256, n)
        a, ra = oracle.ascend(x)
        b, rb = reflibs.ascend(x)
        assert ra == rb and a.tobytes() == b.tobytes()
    for name, nodes in CASES.items():
        a, ra = oracle.ascend(nodes)
        b, rb = reflibs.ascend(nodes)
        assert ra == rb and a.tobytes() == b.tobytes(), name


def test_live_publish_scan_vs_real_node(oracle, reflibs):
    if reflibs is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this box)")
    for name, nodes in CASES.items():
        for kind, inv, sp in [(0, 0, 1), (1, 1, 1), (2, 0, 0), (2, 1, 0), (2, 1, 1)]:
            p = oracle_lib.params(is_new_protocol=int(kind == 2), inverted=inv,
                                  scan_processing=sp, range_max=12.0)
            r, i, m = oracle.publish_scan(nodes, p, 0.2)
            rr, ri, rm = reflibs.publish_scan(nodes, driver_kind=kind, inverted=inv,
                                              scan_processing=sp, range_max=12.0,
                                              scan_duration=0.2)
            assert bytes(m) == bytes(rm), (name, kind, inv, sp)
            assert r.tobytes() == rr.tobytes() and i.tobytes() == ri.tobytes(), (name, kind, inv, sp)


def test_live_publish_scan_fuzz_vs_real_node(oracle, reflibs):
    """Random scans (duplicate angles, any u32 distance, every size class) through the oracle's
    restatement and through the GENUINE RPlidarNode::publish_scan built from /root/reference:
    both run libstdc++'s std::sort with the same comparator, so even the tie order agrees."""
    if reflibs is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this box)")
    rng = np.random.default_rng(11)
    for t in range(400):
        n = int(rng.choice([1, 2, 3, 17, 360, 1000, int(rng.integers(1, 5000))]))
        x = np.zeros(n, oracle_lib.NODE)
        x["angle_z_q14"] = rng.integers(0, 65536, n) if t % 3 else np.sort(rng.integers(0, 65536, n))
        kind_d = t % 4
        if kind_d == 0:
            d = rng.integers(0, 2**32, n, dtype=np.uint64)
        elif kind_d == 1:
            d = rng.integers(0, 200000, n)
        elif kind_d == 2:
            d = rng.choice([0, 1, 599, 600, 47999, 48000, 48001, 4000], n)
        else:
            d = (rng.uniform(0.1, 30.0) * 4000 + rng.normal(0, 40, n)).clip(0, 2**32 - 1)
        x["dist_mm_q2"] = (np.asarray(d, np.uint64) * (rng.random(n) > rng.random())).astype(np.uint32)
        x["quality"] = rng.integers(0, 256, n)
        x["flag"] = rng.integers(0, 4, n)
        kind, inv, sp = int(rng.integers(0, 3)), int(rng.integers(0, 2)), int(rng.integers(0, 2))
        rmax = float(rng.choice([12.0, 40.0, 8.0]))
        dur = float(rng.choice([0.1, 0.0731, 0.2]))
        p = oracle_lib.params(is_new_protocol=int(kind == 2), inverted=inv, scan_processing=sp,
                              range_max=rmax)
        r, i, m = oracle.publish_scan(x, p, dur)
        rr, ri, rm = reflibs.publish_scan(x, driver_kind=kind, inverted=inv, scan_processing=sp,
                                          range_max=rmax, scan_duration=dur)
        assert bytes(m) == bytes(rm), (t, n, kind, inv, sp)
        assert r.tobytes() == rr.tobytes() and i.tobytes() == ri.tobytes(), (t, n, kind, inv, sp)


# ---- extension oracle: internal consistency (parity unpinned, spec = SURVEY §8 a-ext) -----
def test_clip_disabled_reduces_to_reference(oracle):
    nodes = CASES["c1_like_360"]
    a = oracle.publish_scan(nodes, oracle_lib.params(clip_enable=0), 0.1)
    b = oracle.publish_scan(nodes, oracle_lib.params(clip_enable=1, q_min=0, range_min=0.0,
                                                     range_max=1e9), 0.1)
    assert a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes()


def test_cloud_layout_and_voxel_properties(oracle):
    nodes = CASES["ring_8192"]
    p = oracle_lib.params(clip_enable=1, range_max=40.0)
    pts = oracle.scan_to_cloud(nodes, p)
    kept = (nodes["dist_mm_q2"] != 0)
    dm = nodes["dist_mm_q2"].astype(np.float32) / np.float32(4000.0)
    kept &= (dm >= np.float32(0.15)) & (dm <= np.float32(40.0))
    assert pts.shape == (int(kept.sum()), 4) and np.all(pts[:, 2] == 0)
    r = np.hypot(pts[:, 0].astype(np.float64), pts[:, 1].astype(np.float64))
    assert np.max(np.abs(r - dm[kept])) < 1e-5
    vox, cells, counts = oracle.voxel_grid(pts, 0.05)
    assert counts.sum() == len(pts)
    order = np.lexsort((cells[:, 0], cells[:, 1]))
    assert np.array_equal(order, np.arange(len(cells)))  # (iy, ix) ascending
    assert len(np.unique(cells, axis=0)) == len(cells)
    # every centroid lies in (or on the face of) its cell
    leaf = np.float32(0.05)
    assert np.all(np.abs(np.floor(vox[:, 0] / leaf) - cells[:, 0]) <= 1)


def test_ror_mask_small(oracle):
    pts = np.zeros((5, 4), np.float32)
    pts[:, 0] = [0.0, 0.05, 0.09, 1.0, 1.05]
    keep = oracle.ror_mask(pts, 0.10, 2)
    assert keep.tolist() == [True, True, True, False, False]


def numpy_laserscan_to_cloud(ranges, intens, scan_processing, clip=None):
    """Numpy restatement of E7 (second, independent writer of the spec)."""
    count = len(ranges)
    den = np.float64(count) if scan_processing else np.float64(max(count - 1, 1))
    inc = np.float32((2.0 * np.pi) / den)
    keep = np.isfinite(ranges)
    if clip is not None:
        keep &= (ranges >= np.float32(clip[0])) & (ranges <= np.float32(clip[1]))
    theta = (np.arange(count, dtype=np.float32) * inc).astype(np.float32)
    c = np.cos(theta.astype(np.float64)).astype(np.float32)
    s_ = np.sin(theta.astype(np.float64)).astype(np.float32)
    out = np.zeros((int(keep.sum()), 4), np.float32)
    out[:, 0] = (ranges * c)[keep]
    out[:, 1] = (ranges * s_)[keep]
    out[:, 3] = intens[keep]
    return out


@pytest.mark.parametrize("name", ["c1_like_360", "ring_8192", "c2_32000", "kat2", "single_valid"])
def test_laserscan_to_cloud_oracle_vs_numpy(oracle, name):
    nodes = CASES[name]
    for sp in (1, 0):
        p = oracle_lib.params(scan_processing=sp, range_max=40.0)
        r, i, m = oracle.publish_scan(nodes, p, 0.1)
        got = oracle.laserscan_to_cloud(r, i, p)
        want = numpy_laserscan_to_cloud(r, i, sp)
        assert got.shape == want.shape
        # libm vs numpy cos/sin of the same double: both correctly rounded in practice; allow 1 ulp
        assert np.max(np.abs(got.astype(np.float64) - want), initial=0.0) <= 4e-6
        assert np.mean(got.tobytes() == want.tobytes()) == 1 or np.mean(got == want) > 0.999
        pc = oracle_lib.params(scan_processing=sp, range_max=8.0, range_min=0.5, clip_enable=1)
        gotc = oracle.laserscan_to_cloud(r, i, pc)
        assert gotc.shape == numpy_laserscan_to_cloud(r, i, sp, (0.5, 8.0)).shape
    # the projection of a Mode A scan puts every point back on its ring: |p| == range
    p = oracle_lib.params(scan_processing=1, range_max=40.0)
    r, i, m = oracle.publish_scan(CASES["ring_8192"], p, 0.1)
    pts = oracle.laserscan_to_cloud(r, i, p)
    rr = r[np.isfinite(r)]
    assert np.max(np.abs(np.hypot(pts[:, 0].astype(np.float64), pts[:, 1]) - rr)) < 1e-5


def test_fusion_oracle_time_offsets():
    """oracle/fusion_oracle.py, E6 with a start offset (rplgpu_set_scan_time_offsets_dev): properties of the
    spec itself — no offset = the old formula bit for bit; a pure translation moves every point by v * tau
    with tau = t0 + i * dt rounded in that order; at the fused instant itself (tau = 0) a point stays."""
    import sys
    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "oracle"))
    import fusion_oracle as fo
    rng = np.random.default_rng(7)
    n = 4000
    cloud = np.zeros((n, 4), np.float32)
    cloud[:, :2] = rng.uniform(-20, 20, (n, 2)).astype(np.float32)
    cloud[:, 3] = rng.integers(0, 64, n).astype(np.float32)
    idx = np.sort(rng.choice(32000, n, replace=False))
    dt = np.float32(3.125e-6)
    a = fo.deskew_cloud(cloud, idx, (1.2, -0.4, 0.7, dt))
    assert a.tobytes() == fo.deskew_cloud(cloud, idx, (1.2, -0.4, 0.7, dt), None).tobytes()
    t0 = np.float32(-0.0375)
    tr = fo.deskew_cloud(cloud, idx, (2.0, -3.0, 0.0, dt), t0)   # no rotation: x' = x + vx * tau
    tau = (t0 + (idx.astype(np.float32) * dt).astype(np.float32)).astype(np.float32)
    assert tr[:, 0].tobytes() == (cloud[:, 0] + (np.float32(2.0) * tau).astype(np.float32)).astype(np.float32).tobytes()
    assert tr[:, 1].tobytes() == (cloud[:, 1] + (np.float32(-3.0) * tau).astype(np.float32)).astype(np.float32).tobytes()
    assert tr[:, 2:].tobytes() == cloud[:, 2:].tobytes()
    i0 = np.array([12800])                                   # 12800 * 3.125e-6 = 0.04 s, exactly -t0
    pt = cloud[:1]
    at = fo.deskew_cloud(pt, i0, (5.0, 5.0, 1.0, dt), np.float32(-0.04))
    assert np.max(np.abs(at[:, :2] - pt[:, :2])) <= 1e-6     # (0.04f is not 12800 * dt exactly: tau ~ 1e-9 s)
