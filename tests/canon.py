"""Canonical forms + digests for comparing outputs whose order inside equal-angle runs is the
reference's unstable ``std::sort`` (introsort) order.  Shared by the golden generator, the CPU
oracle tests and the GPU parity tests, so that all three compare the same thing."""
from __future__ import annotations

import hashlib

import numpy as np


def digest(arr) -> np.ndarray:
    """SHA-256 of an array's bytes as a uint8[32] array (fits an .npz)."""
    b = np.ascontiguousarray(arr).tobytes()
    return np.frombuffer(hashlib.sha256(b).digest(), np.uint8).copy()


def canon_ascend(nodes: np.ndarray) -> np.ndarray:
    """ascendScanData output with every equal-angle run sorted by the remaining fields."""
    key = np.lexsort((nodes["flag"], nodes["quality"], nodes["dist_mm_q2"], nodes["angle_z_q14"]))
    return nodes[key]


def valid_angles_unique(nodes: np.ndarray) -> bool:
    v = nodes[nodes["dist_mm_q2"] != 0]
    return len(np.unique(v["angle_z_q14"])) == len(v)


def canon_mode_b(nodes: np.ndarray, ranges: np.ndarray, intens: np.ndarray, inverted: int):
    """Mode B output (src/rplidar_node.cpp:663-680) with every equal-angle run sorted by
    (range bits, intensity bits): inside a run the reference's order is introsort's."""
    v = nodes[nodes["dist_mm_q2"] != 0]
    srt = np.sort(v["angle_z_q14"]).astype(np.int64)
    if not inverted:  # idx = count - 1 - i  (:676)
        srt = srt[::-1]
    assert len(srt) == len(ranges) == len(intens)
    if len(srt) == 0:
        return ranges.copy(), intens.copy()
    run = np.r_[0, np.cumsum(np.diff(srt) != 0)]
    rb = np.ascontiguousarray(ranges, np.float32).view(np.uint32)
    ib = np.ascontiguousarray(intens, np.float32).view(np.uint32)
    order = np.lexsort((ib, rb, run))
    return ranges[order], intens[order]


def has_intensity_tie(nodes: np.ndarray, is_new_protocol: int) -> bool:
    """True when two kept samples share (angle, dist_m) but differ in intensity: Mode A's
    winner among them is then introsort's choice upstream."""
    v = nodes[nodes["dist_mm_q2"] != 0]
    if len(v) < 2:
        return False
    dm = (v["dist_mm_q2"].astype(np.float32) / np.float32(4000.0)).view(np.uint32)
    inten = (v["quality"] if is_new_protocol else (v["quality"] >> 2)).astype(np.int64)
    key = (v["angle_z_q14"].astype(np.int64) << 32) | dm.astype(np.int64)
    order = np.argsort(key, kind="stable")
    key, inten = key[order], inten[order]
    starts = np.r_[0, np.flatnonzero(np.diff(key) != 0) + 1]
    return bool(np.any(np.minimum.reduceat(inten, starts) != np.maximum.reduceat(inten, starts)))


def check_large_golden(g, name, nodes, ascend_fn, laserscan_fn):
    """Compare one large case with tests/golden/large_golden.npz (digests of the GENUINE
    reference's outputs).  ``ascend_fn(nodes) -> (out, sl_result)``;
    ``laserscan_fn(nodes, kind, inv, sp) -> (ranges, intens, meta_bytes)``.  Returns the number
    of (case, combination) outputs compared."""
    assert digest(nodes).tobytes() == g[f"{name}__in_sha"].tobytes(), "case generator drifted"
    out, res = ascend_fn(nodes)
    assert int(res) == int(g[f"{name}__asc_res"]), name
    if int(res) == 0:
        assert digest(canon_ascend(out)).tobytes() == g[f"{name}__asc_sha"].tobytes(), name
    checked = 0
    for kind in (0, 1, 2):
        for inv in (0, 1):
            for sp in (0, 1):
                tag = f"{name}__k{kind}_i{inv}_s{sp}"
                r, i, mb = laserscan_fn(nodes, kind, inv, sp)
                assert mb == g[tag + "__meta"].tobytes(), tag
                if sp:
                    assert digest(r).tobytes() == g[tag + "__ranges_sha"].tobytes(), tag
                    if tag + "__intens_sha" in g:
                        assert digest(i).tobytes() == g[tag + "__intens_sha"].tobytes(), tag
                else:
                    cr, ci = canon_mode_b(nodes, r, i, inv)
                    assert digest(cr).tobytes() == g[tag + "__ranges_sha"].tobytes(), tag
                    assert digest(ci).tobytes() == g[tag + "__intens_sha"].tobytes(), tag
                checked += 1
    return checked
