"""CPU tests for SURVEY.md §8(f) rows 1-2: pin the restated unpackers / scan assembly
(oracle/oracle_unpack.cpp) against
  (1) tests/golden/unpack_golden.npz — outputs of the GENUINE reference unpackers
      (src/sdk/src/dataunpacker/unpacker/handler_*.cpp) and ScanDataHolder
      (src/sdk/src/sl_lidar_driver.cpp:236-360), produced by tests/golden/make_golden.py through
      oracle/_ref/libunpackref.so;
  (2) the genuine code itself on fresh random / corrupted streams when oracle/_ref is built
      (build container only).
Integer work: everything must match bit for bit."""
from pathlib import Path

import numpy as np
import pytest

from rplidar_ros2_driver_amd import capsules as cp
from tests import oracle_lib

GOLD = Path(__file__).resolve().parent / "golden" / "unpack_golden.npz"
ALL_ANS = (0x81, 0x82, 0x83, 0x84, 0x85, 0x86)


def golden_cases():
    g = np.load(GOLD)
    tags = sorted({k.split("__")[0] for k in g.files})
    return g, tags


def test_generator_is_stable():
    """The committed golden inputs are what the generator still produces."""
    g, tags = golden_cases()
    for t in tags:
        ans, nf, seed, corrupt, dur, _ = (int(v) for v in g[t + "__meta"])
        data = cp.make_stream(ans, nf, seed, corrupt=bool(corrupt), payload=str(g[t + "__payload"]),
                              frames_per_rev=float(g[t + "__fpr"]))
        assert data.tobytes() == g[t + "__bytes"].tobytes(), t


def test_unpack_matches_genuine_golden(oracle):
    g, tags = golden_cases()
    assert len(tags) >= 12
    seen = set()
    for t in tags:
        ans, nf, seed, corrupt, dur, dense_last = (int(v) for v in g[t + "__meta"])
        nodes, rst, err, _ = oracle.unpack(ans, g[t + "__bytes"], dur, state=(dense_last, 0))
        assert nodes.tobytes() == g[t + "__nodes"].tobytes(), t
        assert list(rst) == list(g[t + "__reset_at"]), t
        assert err == int(g[t + "__n_err"]), t
        seen.add(ans)
    assert seen == set(ALL_ANS)


def test_framing_plus_decoding_equals_unpack(oracle):
    """The two-stage split the product uses (host framer -> framed decode) is the same function."""
    g, tags = golden_cases()
    for t in tags:
        ans, nf, seed, corrupt, dur, dense_last = (int(v) for v in g[t + "__meta"])
        data = g[t + "__bytes"]
        off, gap = oracle.frame_stream(ans, data)
        nodes, rst, err, _ = oracle.unpack_frames(ans, data, off, gap, dur, state=(dense_last, 0))
        assert nodes.tobytes() == g[t + "__nodes"].tobytes(), t
        assert list(rst) == list(g[t + "__reset_at"]), t
        if not corrupt:  # a clean stream is exactly the frames back to back
            S = cp.FRAME_SIZE[ans]
            assert list(off) == list(range(0, nf * S, S)) and not gap.any()


def test_segment_matches_genuine_golden(oracle):
    g, tags = golden_cases()
    nscans = 0
    for t in tags:
        for cap in (8192, 37):
            so, offs = oracle.segment(g[t + "__nodes"], g[t + "__reset_at"], cap)
            assert so.tobytes() == g[t + f"__scans{cap}"].tobytes(), (t, cap)
            assert list(offs) == list(g[t + f"__scan_off{cap}"]), (t, cap)
            nscans += len(offs) - 1
    assert nscans > 50


def test_segment_rules(oracle):
    """ScanDataHolder rules spelled out (src/sdk/src/sl_lidar_driver.cpp:272-310)."""
    n = np.zeros(12, oracle_lib.NODE)
    n["dist_mm_q2"] = np.arange(12) + 1
    n["flag"] = [2, 2, 1, 2, 2, 1, 2, 1, 2, 2, 2, 1]  # nodes 0,1 precede the first sync
    so, offs = oracle.segment(n, np.zeros(0, np.uint32), 8192)
    assert list(offs) == [0, 3, 5, 9]  # [2,3,4] [5,6] [7,8,9,10]; the scan opened by node 11 is not complete
    assert list(so["dist_mm_q2"]) == [3, 4, 5, 6, 7, 8, 9, 10, 11]
    # a rewind before node 4 drops nodes 2,3 and everything up to the next sync node
    so, offs = oracle.segment(n, np.array([4], np.uint32), 8192)
    assert list(so["dist_mm_q2"]) == [6, 7, 8, 9, 10, 11] and list(offs) == [0, 2, 6]
    # max_count 2: the last slot keeps being overwritten
    so, offs = oracle.segment(n, np.zeros(0, np.uint32), 2)
    assert list(so["dist_mm_q2"]) == [3, 5, 6, 7, 8, 11]


@pytest.mark.parametrize("ans", ALL_ANS)
def test_unpack_matches_genuine_live(oracle, ans):
    ref = oracle_lib.load_ref_unpack()
    if ref is None:
        pytest.skip("oracle/_ref/libunpackref.so not built (no /root/reference here)")
    total = 0
    for seed in range(8):
        for corrupt in (False, True):
            fpr = [12.3, 3.1, 40.0, 7.7, 300.0, 1.5, 12.3, 90.0][seed]
            dur = [125, 125, 32, 20, 2, 500, 1000000, 1][seed]
            data = cp.make_stream(ans, 50 if ans != 0x81 else 600, 100 + seed, corrupt=corrupt,
                                  payload="random" if seed % 2 else "ring", frames_per_rev=fpr)
            # (the genuine dense decoder keeps a process-wide static: `dense_last` follows it)
            r_nodes, r_rst, r_err = ref.unpack(ans, data, dur, chunk=[0, 1, 7, 84, 1000, 33, 5, 2][seed])
            last = test_unpack_matches_genuine_live.dense_last if ans == 0x85 else 0
            o_nodes, o_rst, o_err, st = oracle.unpack(ans, data, dur, state=(last, 0))
            if ans == 0x85:
                test_unpack_matches_genuine_live.dense_last = st[0]
            assert o_nodes.tobytes() == r_nodes.tobytes(), (seed, corrupt)
            assert list(o_rst) == list(r_rst) and o_err == r_err
            so, offs = oracle.segment(o_nodes, o_rst, 64)
            rso, roffs = ref.segment(o_nodes, o_rst, 64)
            assert so.tobytes() == rso.tobytes() and list(offs) == list(roffs)
            total += len(r_nodes)
    assert total > 1000


test_unpack_matches_genuine_live.dense_last = 0


def test_unpack_fuzz_vs_genuine_live(oracle):
    """Random lengths, heavy fault rates and pure byte soup through the oracle's restatement and
    through the GENUINE SDK unpackers + ScanDataHolder (fed in random chunk sizes)."""
    ref = oracle_lib.load_ref_unpack()
    if ref is None:
        pytest.skip("oracle/_ref/libunpackref.so not built (no /root/reference here)")
    rng = np.random.default_rng(4242)
    for t in range(240):
        ans = ALL_ANS[t % len(ALL_ANS)]
        nf = int(rng.choice([1, 2, 3, 17, 64, int(rng.integers(1, 200))])) * (8 if ans == 0x81 else 1)
        dur = int(rng.choice([125, 32, 20, 2, 1000000, 476]))
        mode = int(rng.integers(0, 4))
        frames = cp.make_frames(ans, nf, int(rng.integers(0, 1 << 30)),
                                payload=str(rng.choice(["random", "ring"])),
                                frames_per_rev=float(rng.choice([1.5, 3.1, 12.3, 40.0, 300.0])),
                                first_sync=bool(rng.integers(0, 2)))
        if mode == 0:
            data = frames.reshape(-1)
        elif mode == 1:
            data = cp.corrupt_stream(ans, frames, int(rng.integers(0, 1 << 30)))
        elif mode == 2:
            data = cp.corrupt_stream(ans, frames, int(rng.integers(0, 1 << 30)), p_checksum=0.3,
                                     p_sync=0.2, p_garbage=0.2, p_revstart=0.2, p_jump=0.2)
        else:
            data = rng.integers(0, 256, int(rng.integers(0, 3000)), dtype=np.uint8)
        chunk = int(rng.choice([0, 1, 7, 84, 1000, 33]))
        r_nodes, r_rst, r_err = ref.unpack(ans, data, dur, chunk=chunk)
        last = test_unpack_matches_genuine_live.dense_last if ans == 0x85 else 0
        o_nodes, o_rst, o_err, st = oracle.unpack(ans, data, dur, state=(last, 0))
        if ans == 0x85:
            test_unpack_matches_genuine_live.dense_last = st[0]
        ctx = (t, hex(ans), nf, mode, chunk)
        assert o_nodes.tobytes() == r_nodes.tobytes(), ctx
        assert list(o_rst) == list(r_rst) and o_err == r_err, ctx
        mc = int(rng.choice([64, 8192, 5]))
        so, offs = oracle.segment(o_nodes, o_rst, mc)
        rso, roffs = ref.segment(o_nodes, o_rst, mc)
        assert so.tobytes() == rso.tobytes() and list(offs) == list(roffs), ctx
