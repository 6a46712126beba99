"""SECOND, INDEPENDENT WRITER of the extension spec E1 / E2 / E4 / E5 (SURVEY.md §8 a-ext).

TEST INFRASTRUCTURE.  Only tests/ and tests/golden/make_ext_golden.py import this file; the product
package never does.

Why it exists: nothing in the reference implements the north-star's PointCloud2 extensions, so the
C++ extension oracle (oracle/oracle.cpp: orc_scan_to_cloud, orc_voxel_grid, orc_ror_mask) cannot be
pinned to a reference-held vector — "parity unpinned".  A kernel and an oracle written by one hand
could share one misreading of the spec.  This file is a second reading, written from the TEXT of
SURVEY.md §8(a-ext) and from the reference's own per-sample expressions
(/root/reference/src/rplidar_node.cpp:584-599 for the mask, angle, range and intensity of a sample,
:646-651 for the invert rule), NOT from oracle.cpp, and with different machinery on purpose:

  * whole-array numpy float32 / float64 expressions instead of per-sample loops;
  * a (cos, sin) look-up table over all 65 536 angle words built with Python's math.cos / math.sin
    (the spec's "host-built float LUT indexed by angle_z_q14");
  * the voxel grid by np.unique over (iy, ix) pairs + np.add.at in float64 (sequential, i.e. in
    ascending sample order) instead of a comparison sort of tags;
  * radius-outlier removal by scipy's cKDTree for the candidate pairs + the exact float32
    products-then-sum predicate on the candidates, instead of the O(n^2) double loop.

The two writers are compared on every case of tests/cases.py and on bench-regime scans in
tests/test_ext_second_writer.py (cells, counts and the keep masks exact, intensity bit for bit,
x / y within 1e-6 m); tests/golden/ext_golden.npz holds THIS file's outputs, so the C++ oracle (and
through it the kernels) is also held to a committed vector it did not produce.  Still "unpinned" in
the sense of the task: no reference-held vector exists for this path.
"""
from __future__ import annotations

import math

import numpy as np

F32 = np.float32
F64 = np.float64
TWO_PI = 2.0 * math.pi


# ---- per-sample quantities, rplidar_node.cpp:584-599 --------------------------------------------
def angle_rad_table() -> np.ndarray:
    """angle_rad of every angle word (:588-589, with the unreachable wraps of :594-599 kept)."""
    q = np.arange(65536, dtype=np.int64)
    deg = (q.astype(F32) * F32(90.0)) / F32(16384.0)               # float * float / float
    k = F64(math.pi) / F64(F32(180.0))                              # (M_PI / 180.0f): a double
    rad = (deg.astype(F64) * k).astype(F32)                         # float = float * double
    lo = rad < F32(0.0)
    rad = np.where(lo, (rad.astype(F64) + F64(F32(2.0)) * math.pi).astype(F32), rad)
    hi = rad.astype(F64) >= F64(F32(2.0)) * math.pi
    rad = np.where(hi, (rad.astype(F64) - F64(F32(2.0)) * math.pi).astype(F32), rad)
    return rad


def invert(rad: np.ndarray) -> np.ndarray:
    """:646-651 — angle = 2 pi - angle (double arithmetic, stored to float), >= 2 pi -> -= 2 pi."""
    a = (F64(F32(2.0)) * math.pi - rad.astype(F64)).astype(F32)
    hi = a.astype(F64) >= F64(F32(2.0)) * math.pi
    return np.where(hi, (a.astype(F64) - F64(F32(2.0)) * math.pi).astype(F32), a)


_LUT = {}


def cos_sin_lut(inverted: bool):
    """E2: (float)cos((double)theta), (float)sin((double)theta) for every angle word."""
    if inverted not in _LUT:
        th = angle_rad_table()
        if inverted:
            th = invert(th)
        c = np.array([math.cos(float(t)) for t in th], F64).astype(F32)
        s = np.array([math.sin(float(t)) for t in th], F64).astype(F32)
        _LUT[inverted] = (c, s)
    return _LUT[inverted]


def dist_m(nodes) -> np.ndarray:
    return nodes["dist_mm_q2"].astype(F32) / F32(4000.0)           # :590 (u32 -> f32, RNE)


def intensity(nodes, is_new_protocol: bool) -> np.ndarray:
    q = nodes["quality"].astype(np.uint32)
    return (q if is_new_protocol else (q >> 2)).astype(F32)         # :591-592


# ---- E1 ------------------------------------------------------------------------------------------
def keep_mask(nodes, *, clip_enable: bool, q_min: int, range_min: float, range_max: float) -> np.ndarray:
    """keep = dist != 0 [&& quality >= q_min && range_min <= dist_m <= range_max when clipping]."""
    keep = nodes["dist_mm_q2"] != 0
    if clip_enable:
        dm = dist_m(nodes)
        keep = keep & (nodes["quality"].astype(np.uint32) >= np.uint32(q_min))
        keep = keep & (dm >= F32(range_min)) & (dm <= F32(range_max))
    return keep


# ---- E2 (+ E3's point layout: x, y, z, intensity as four float32) ----------------------------------
def scan_to_points(nodes, *, is_new_protocol=False, inverted=False, clip_enable=False, q_min=0,
                   range_min=0.15, range_max=12.0) -> np.ndarray:
    keep = keep_mask(nodes, clip_enable=clip_enable, q_min=q_min, range_min=range_min, range_max=range_max)
    c, s = cos_sin_lut(bool(inverted))
    q = nodes["angle_z_q14"].astype(np.int64)[keep]
    dm = dist_m(nodes)[keep]
    pts = np.zeros((int(keep.sum()), 4), F32)
    pts[:, 0] = dm * c[q]
    pts[:, 1] = dm * s[q]
    pts[:, 3] = intensity(nodes, bool(is_new_protocol))[keep]
    return pts


# ---- E5 ------------------------------------------------------------------------------------------
def ror_keep(pts: np.ndarray, radius: float, k: int) -> np.ndarray:
    """keep i iff #{j != i : (xi-xj)^2 + (yi-yj)^2 <= r^2} >= k, float32, products then sum."""
    from scipy.spatial import cKDTree

    n = len(pts)
    if n == 0:
        return np.zeros(0, bool)
    r = float(F32(radius))
    r2 = F32(radius) * F32(radius)
    xy = pts[:, :2].astype(F64)                                      # exact
    tree = cKDTree(xy)
    # A neighbour nearer than r - eps passes the float32 predicate and one farther than r + eps
    # fails it whatever the rounding (the float32 evaluation of dx^2 + dy^2 is within a few 1e-7
    # relative of the true value; eps = 1e-6 m is 1e-5 relative at r = 0.1 m).  Only points with a
    # neighbour inside that thin shell are re-counted with the exact float32 arithmetic.
    eps = 1e-6
    n_in = tree.query_ball_point(xy, max(r - eps, 0.0), return_length=True)
    n_out = tree.query_ball_point(xy, r + eps, return_length=True)
    cnt = n_in.astype(np.int64) - 1                                  # the point itself
    for i in np.nonzero(n_in != n_out)[0]:
        nb = np.asarray(tree.query_ball_point(xy[i], r + eps), np.int64)
        nb = nb[nb != i]
        dx = pts[i, 0] - pts[nb, 0]
        dy = pts[i, 1] - pts[nb, 1]
        d2 = dx * dx + dy * dy                                       # float32: products, then the sum
        cnt[i] = int(np.count_nonzero(d2 <= r2))
    return cnt >= int(k)


# ---- E4 ------------------------------------------------------------------------------------------
def voxel_grid(pts: np.ndarray, leaf: float):
    """One point per occupied cell: cell = (floor(x / leaf), floor(y / leaf)) in float32 then int32;
    centroid and mean intensity = float64 sums in ascending sample order / count, rounded to float32;
    cells in (iy, ix) ascending order.  Returns (points, cells[ix, iy], counts)."""
    if len(pts) == 0:
        return np.zeros((0, 4), F32), np.zeros((0, 2), np.int32), np.zeros(0, np.uint32)
    lf = F32(leaf)
    ix = np.floor(pts[:, 0] / lf).astype(np.int32)
    iy = np.floor(pts[:, 1] / lf).astype(np.int32)
    key = (iy.astype(np.int64) << 32) + (ix.astype(np.int64) + (1 << 31))   # (iy, ix) order
    uniq, inv, counts = np.unique(key, return_inverse=True, return_counts=True)
    sums = np.zeros((len(uniq), 4), F64)
    np.add.at(sums, inv, pts.astype(F64))                            # sequential: sample order
    out = (sums / counts[:, None].astype(F64)).astype(F32)
    cix = ((uniq & 0xFFFFFFFF) - (1 << 31)).astype(np.int32)
    ciy = (uniq >> 32).astype(np.int32)
    return out, np.stack([cix, ciy], axis=1), counts.astype(np.uint32)


# ---- the pipeline E1 -> E2 -> (E5) -> (E4) ------------------------------------------------------------
def cloud_pipeline(nodes, *, is_new_protocol=False, inverted=False, clip_enable=False, q_min=0,
                   range_min=0.15, range_max=12.0, ror_enable=False, ror_radius=0.10,
                   ror_min_neighbors=2, voxel_enable=False, voxel_leaf=0.05):
    pts = scan_to_points(nodes, is_new_protocol=is_new_protocol, inverted=inverted,
                         clip_enable=clip_enable, q_min=q_min, range_min=range_min, range_max=range_max)
    if ror_enable and len(pts):
        pts = pts[ror_keep(pts, ror_radius, ror_min_neighbors)]
    if not voxel_enable:
        return pts, None, None
    return voxel_grid(pts, voxel_leaf)
