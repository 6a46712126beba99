"""The extension spec E1 / E2 / E4 / E5 (SURVEY.md §8 a-ext) has no reference implementation, so
its C++ oracle (oracle/oracle.cpp) cannot be pinned to reference-held vectors.  These tests hold it
to a SECOND, independent writer of the same spec text (oracle/ext_second_writer.py: numpy / scipy,
np.unique + np.add.at voxel grid, cKDTree radius search) and to that writer's committed outputs
(tests/golden/ext_golden.npz, tests/golden/make_ext_golden.py):

  * every case of tests/cases.py under eight parameter sets (clip, invert, protocol, quality
    threshold, two leaf sizes, ROR before the grid);
  * 64 scans of each bench regime (clean ring, 1 cm range noise, uniform random; 32 000 samples),
    and the C5 regime (noisy ring + ROR + voxel) on 8 scans.

Bar: point counts, cell indices, per-cell counts and the ROR keep decisions exact; intensity bit for
bit; x / y within 1e-6 m (in fact they come out bit-identical: both writers take cos / sin of the
same double from libm and sum in float64 in sample order).  Still "parity unpinned" for the a-ext
row — two independent writers, no reference vector."""
import numpy as np
import pytest

from oracle import ext_second_writer as sw
from rplidar_ros2_driver_amd import synth
from tests import oracle_lib
from tests.cases import CASES
from tests.golden.make_ext_golden import EXT_PARAM_SETS, FULL_MAX_POINTS, ROR_MAX_N, digest

GOLD = np.load(oracle_lib.ROOT / "tests" / "golden" / "ext_golden.npz")


def orc_params(kw):
    return oracle_lib.params(**{k: (int(v) if isinstance(v, bool) else v) for k, v in kw.items()})


def compare(name, got, want):
    (gp, gc, gn), (wp, wc, wn) = got, want
    assert gp.shape == wp.shape, (name, gp.shape, wp.shape)
    if wc is not None:
        assert np.array_equal(gc, wc), name          # cell indices, (iy, ix) order
        assert np.array_equal(gn, wn), name          # samples per cell
    assert gp[:, 3].tobytes() == wp[:, 3].tobytes(), name   # intensity, bit for bit
    assert np.all(gp[:, 2] == 0) and np.all(wp[:, 2] == 0), name
    if len(gp):
        assert np.max(np.abs(gp[:, :2].astype(np.float64) - wp[:, :2])) <= 1e-6, name
    return gp.tobytes() == wp.tobytes()


@pytest.mark.parametrize("tag,kw", EXT_PARAM_SETS, ids=[t for t, _ in EXT_PARAM_SETS])
def test_cpp_oracle_vs_second_writer_on_all_cases(oracle, tag, kw):
    exact = total = 0
    for name, nodes in CASES.items():
        if kw.get("ror_enable") and len(nodes) > ROR_MAX_N:
            continue  # (the C++ oracle's ROR is the O(n^2) definition)
        got = oracle.cloud_pipeline(nodes, orc_params(kw))
        if not kw.get("voxel_enable"):
            got = (got[0], None, None)
        want = sw.cloud_pipeline(nodes, **kw)
        exact += compare(f"{name}/{tag}", got, want)
        total += 1
        # ... and against the committed vectors of the second writer
        key = f"{name}__{tag}"
        assert int(GOLD[key + "__n"]) == len(got[0]), key
        assert GOLD[key + "__sha"].tobytes() == digest(*got).tobytes(), key
        if len(want[0]) <= FULL_MAX_POINTS:
            assert GOLD[key + "__pts"].tobytes() == got[0].tobytes(), key
            if want[1] is not None:
                assert np.array_equal(GOLD[key + "__cells"], got[1]), key
                assert np.array_equal(GOLD[key + "__counts"], got[2]), key
    assert total > 30
    assert exact == total, f"{tag}: {total - exact} of {total} cases agree within 1e-6 m but not bit for bit"


REGIMES = {
    "ring_clean": {},
    "ring_noise_1cm": {"noise_m": 0.01},
    "uniform": {"kind": "uniform"},
}


@pytest.mark.parametrize("regime", sorted(REGIMES))
def test_bench_regimes_voxel(oracle, regime):
    """64 scans of the bench batch (seed 2026, as bench.py) per regime: E1 + E2 + E4."""
    kw = dict(clip_enable=True, q_min=0, range_min=0.15, range_max=40.0, voxel_enable=True, voxel_leaf=0.05)
    batch = synth.make_batch(2026, 64, 32000, **REGIMES[regime])
    for b in range(len(batch)):
        got = oracle.cloud_pipeline(batch[b], orc_params(kw))
        want = sw.cloud_pipeline(batch[b], **kw)
        assert compare(f"{regime}[{b}]", got, want)


def test_bench_regime_c5_ror_voxel(oracle):
    """The C5 regime (1 cm noise, ROR r = 0.10 m k = 2, then the 5 cm grid) on 8 scans of 8 192
    samples (the C++ oracle's ROR is quadratic) and the ROR decisions alone on one 32 000-sample
    scan of a SMALL ring, where a point has ~1000 neighbours."""
    kw = dict(clip_enable=True, range_max=40.0, ror_enable=True, ror_radius=0.10, ror_min_neighbors=2,
              voxel_enable=True, voxel_leaf=0.05)
    batch = synth.make_batch(2027, 8, 8192, noise_m=0.01)
    for b in range(len(batch)):
        assert compare(f"c5[{b}]", oracle.cloud_pipeline(batch[b], orc_params(kw)),
                       sw.cloud_pipeline(batch[b], **kw))
    # sparse data: most points are outliers, the decision is not trivially "keep"
    sparse = synth.make_scan(2027, 100, 4096, kind="uniform")
    pts = oracle.scan_to_cloud(sparse, orc_params(dict(clip_enable=True, range_max=40.0)))
    for r, k in ((0.10, 2), (0.25, 1), (0.5, 3)):
        a, b = oracle.ror_mask(pts, r, k), sw.ror_keep(pts, r, k)
        assert np.array_equal(a, b) and 0 < a.sum() < len(a), (r, k, int(a.sum()))
    small = synth.make_scan(2027, 101, 32000, r0_range=(1.0, 1.2))
    pts = oracle.scan_to_cloud(small, orc_params(dict(clip_enable=True, range_max=40.0)))
    assert np.array_equal(oracle.ror_mask(pts, 0.10, 2), sw.ror_keep(pts, 0.10, 2))


def test_second_writer_angle_table_against_the_spec_examples():
    """SURVEY.md §8(c) KAT-2's constants: max angle_rad for q14 = 65535 is 6.28308964f < 2 pi_f, and
    the inverted angle of word 0 is 1.7484555e-07 (float(2 pi) as a double minus 0, stored to float,
    then >= 2 pi -> -= 2 pi in double)."""
    th = sw.angle_rad_table()
    assert th[65535] == np.float32(6.28308964) and th[0] == 0
    assert sw.invert(th[:1])[0] == np.float32(1.7484555e-07)
    assert np.all(np.diff(th.astype(np.float64)) > 0)
