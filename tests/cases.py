"""Shared raw-scan test inputs: the edge cases of SURVEY.md §8(d) plus seeded synthetic
scans at the BASELINE shapes (360 / 8192 / 32000 / 32768 samples)."""
import numpy as np

from rplidar_ros2_driver_amd import NODE_DTYPE, synth


def edge_scans():
    """Edge cases of SURVEY.md §8(d)."""
    rng = np.random.default_rng(7)
    out = {}

    def mk(n):
        return np.zeros(n, NODE_DTYPE)

    a = mk(64)
    a["angle_z_q14"] = np.arange(64) * 1000
    out["all_invalid"] = a
    b = a.copy()
    b["dist_mm_q2"][37] = 8000
    b["quality"][37] = 88
    out["single_valid"] = b
    c = mk(1)
    c["dist_mm_q2"] = 4000
    c["angle_z_q14"] = 123
    out["n1_valid"] = c
    out["n1_invalid"] = mk(1)
    d = mk(2)
    d["angle_z_q14"] = [40000, 100]
    d["dist_mm_q2"] = [0, 5000]
    out["n2_lead_invalid"] = d
    e = synth.make_scan(3, 0, 360, invalid_p=0.0)
    e["dist_mm_q2"][:25] = 0
    e["dist_mm_q2"][-30:] = 0
    out["lead_trail_runs"] = e
    f = synth.make_scan(3, 1, 500, invalid_p=0.2)
    f["angle_z_q14"] = np.sort(rng.integers(0, 200, 500)).astype(np.uint16) * 300  # many duplicates
    out["dup_angles"] = f
    g = synth.make_scan(3, 2, 300)
    g["angle_z_q14"][-1] = 65535
    g["angle_z_q14"][0] = 0
    g["dist_mm_q2"][0] = 7000
    g["dist_mm_q2"][-1] = 9000
    out["q14_extremes"] = g
    h = mk(3)  # KAT-2 of SURVEY.md §8(c)
    h["angle_z_q14"] = [0, 65535, 32768]
    h["dist_mm_q2"] = [4000, 8000, 4]
    h["quality"] = [200, 100, 4]
    out["kat2"] = h
    k = synth.make_scan(3, 3, 777, kind="uniform", jitter=400, rotate=True)
    out["unsorted_uniform"] = k
    m = mk(100)  # huge distances: u32 -> f32 rounding and equal dist_m from different dist_q2
    m["angle_z_q14"] = (np.arange(100) // 2) * 600
    m["dist_mm_q2"] = 0xFFFFFF00 + (np.arange(100) % 7)
    m["quality"] = np.arange(100)
    out["huge_dist"] = m
    return out


def gen_cases():
    cases = dict(edge_scans())
    cases["c1_like_360"] = synth.make_scan(11, 0, 360, invalid_p=0.05)
    cases["ring_8192"] = synth.make_scan(11, 1, 8192)
    cases["ring_8192_rot_jit"] = synth.make_scan(11, 2, 8192, jitter=30, rotate=True)
    cases["c2_32000"] = synth.make_scan(11, 3, 32000)
    cases["c2_32000_newproto"] = synth.make_scan(11, 4, 32000, new_protocol=True, jitter=3)
    cases["full_32768_unsorted"] = synth.make_scan(11, 5, 32768, kind="uniform", jitter=2000,
                                                   rotate=True, invalid_p=0.3)
    cases.update(random_scans())
    cases.update(duplicate_angle_scans())
    return cases


def duplicate_angle_scans():
    """Large scans whose angle words repeat — what real scans look like: ascendScanData's fill pass
    (src/sdk/src/sl_lidar_driver.cpp:171-178) gives every invalid node an interpolated angle that
    often lands on a neighbour's word, and a fast-spinning sensor repeats words by itself.  Inside a
    run of equal angles the reference's order is its unstable std::sort's; the golden digests
    (tests/golden/large_golden.npz) therefore pin the MULTISET of every run — tests/canon.py —
    for ascendScanData and Mode B, the ranges for Mode A, and Mode A's intensities wherever no
    two tied samples differ in intensity.  These cases make that contract a test."""
    out = {}
    rng = np.random.default_rng(20260924)
    a = synth.make_scan(13, 0, 8192, invalid_p=0.15)
    a["angle_z_q14"] = (np.sort(rng.integers(0, 2048, 8192)) * 32).astype(np.uint16)  # ~4 per word
    out["dup_angles_8192"] = a
    b = synth.make_scan(13, 1, 32000, invalid_p=0.25, jitter=3)
    b["angle_z_q14"] = (b["angle_z_q14"].astype(np.uint32) & 0xFFF0).astype(np.uint16)  # runs of ~8
    out["dup_angles_32000"] = b
    c = synth.make_scan(13, 2, 12000, kind="uniform", invalid_p=0.3, rotate=True)
    c["angle_z_q14"] = rng.integers(0, 3000, 12000).astype(np.uint16) * 21  # unsorted, heavy ties
    c["quality"] = (rng.integers(0, 4, 12000) * 64).astype(np.uint8)      # and few intensity values
    out["dup_angles_unsorted_12000"] = c
    return out


def random_scans():
    """Seeded random scans small enough for the golden fixtures: unique angle words (so the
    reference's unstable sorts leave no choice), in sorted / rotated / shuffled order, with
    distances from several awkward families."""
    out = {}
    rng = np.random.default_rng(20260923)
    for k in range(16):
        n = int(rng.choice([1, 5, 33, 64, 65, 129, 200, 300]))
        m = np.zeros(n, NODE_DTYPE)
        q = rng.choice(65536, size=n, replace=False)
        if k % 3 == 0:
            q = np.sort(q)
        elif k % 3 == 1:
            q = np.roll(np.sort(q), int(rng.integers(0, n)))
        m["angle_z_q14"] = q
        fam = k % 4
        if fam == 0:
            d = rng.integers(0, 2**32, n, dtype=np.uint64)
        elif fam == 1:
            d = rng.integers(600, 160001, n)
        elif fam == 2:
            d = rng.choice([1, 599, 600, 601, 47999, 48000, 48001, 159999, 160000, 160001], n)
        else:
            d = (rng.uniform(0.2, 30.0) * 4000 + rng.normal(0, 40, n)).clip(1, 2**32 - 1)
        d = np.asarray(d, np.uint64)
        d[rng.random(n) < [0.0, 0.1, 0.5, 0.9][k % 4]] = 0
        m["dist_mm_q2"] = d.astype(np.uint32)
        m["quality"] = rng.integers(0, 256, n)
        m["flag"] = rng.integers(0, 4, n)
        out[f"random_{k:02d}"] = m
    return out


CASES = gen_cases()

# cases small enough to ship as golden fixtures (inputs + genuine-reference outputs)
GOLDEN_CASES = [k for k, v in CASES.items() if len(v) <= 1000]
# the large cases (8192 ... 32768 samples, config 2's 32 000 among them) ship as SHA-256 digests
# of the genuine reference's outputs (tests/golden/large_golden.npz, canonical forms of
# tests/canon.py where the reference's order is introsort's): bit-exact work needs no more
LARGE_GOLDEN_CASES = [k for k, v in CASES.items() if len(v) > 1000]
