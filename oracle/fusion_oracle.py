"""Restatement of the multi-sensor fusion step (first step).

TEST INFRASTRUCTURE ONLY — imported by tests/ and nothing else.

PARITY UNPINNED: the reference has no fusion code; it broadcasts one static identity transform
base_link -> frame_id per node (src/rplidar_node.cpp:183-197) and nothing consumes it.  The spec
is fixed here (and in include/rplgpu_msg.h): a cloud point (x, y, z, intensity) of scan b with
pose M_b = [R | t] (row-major 3x4 float32) becomes

    x' = ((r00*x + r01*y) + r02*z) + t0        (every operation rounded to float32,
    y' = ((r10*x + r11*y) + r12*z) + t1         products first, sums left to right, no FMA)
    z' = ((r20*x + r21*y) + r22*z) + t2
    intensity unchanged,

and the fused cloud is the concatenation of the transformed clouds.
"""
from __future__ import annotations

import numpy as np


def transform_cloud(xyzi: np.ndarray, pose: np.ndarray) -> np.ndarray:
    """xyzi (n, 4) float32, pose (3, 4) or (12,) float32 -> (n, 4) float32."""
    p = np.asarray(xyzi, np.float32).reshape(-1, 4)
    m = np.asarray(pose, np.float32).reshape(3, 4)
    out = np.empty_like(p)
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    for r in range(3):  # numpy float32 arithmetic rounds after every operation: no FMA
        a = (m[r, 0] * x).astype(np.float32)
        b = (m[r, 1] * y).astype(np.float32)
        c = (m[r, 2] * z).astype(np.float32)
        s = (a + b).astype(np.float32)
        s = (s + c).astype(np.float32)
        out[:, r] = (s + m[r, 3]).astype(np.float32)
    out[:, 3] = p[:, 3]
    return out


def planar_pose(yaw_rad: float, tx: float, ty: float, tz: float = 0.0) -> np.ndarray:
    """A 2-D lidar mounted flat: rotation about z + translation, as a (3, 4) float32 [R | t]."""
    c, s = np.float32(np.cos(yaw_rad)), np.float32(np.sin(yaw_rad))
    return np.array([[c, -s, 0, tx], [s, c, 0, ty], [0, 0, 1, tz]], np.float32)


def deskew_cloud(xyzi: np.ndarray, sample_index: np.ndarray, motion, t0=None) -> np.ndarray:
    """E6 motion de-skew (include/rplgpu_msg.h): xyzi (n, 4) float32 = the plain cloud,
    sample_index (n,) = input index of every kept sample, motion = (vx, vy, wz, time_increment).
    t0 (rplgpu_set_scan_time_offsets_dev): time of the scan's first sample relative to the instant
    the points are wanted at, tau = t0 + float(i) * time_increment; None: tau = float(i) * time_increment.
    Every operation rounded to float32, in the documented order."""
    f = np.float32
    p = np.asarray(xyzi, np.float32).reshape(-1, 4)
    vx, vy, wz, dt = (f(v) for v in motion)
    tau = (np.asarray(sample_index).astype(np.float32) * dt).astype(np.float32)
    if t0 is not None:
        tau = (f(t0) + tau).astype(np.float32)
    a = (wz * tau).astype(np.float32)
    a2 = (a * a).astype(np.float32)
    ts = (a2 * (f(1.0) / f(120.0))).astype(np.float32)
    ts = (ts + (f(-1.0) / f(6.0))).astype(np.float32)
    ts = (a2 * ts).astype(np.float32)
    ts = (ts + f(1.0)).astype(np.float32)
    sn = (a * ts).astype(np.float32)
    tc = (a2 * (f(-1.0) / f(720.0))).astype(np.float32)
    tc = (tc + (f(1.0) / f(24.0))).astype(np.float32)
    tc = (a2 * tc).astype(np.float32)
    tc = (tc + f(-0.5)).astype(np.float32)
    tc = (a2 * tc).astype(np.float32)
    cn = (tc + f(1.0)).astype(np.float32)
    x, y = p[:, 0], p[:, 1]
    out = p.copy()
    out[:, 0] = (((cn * x).astype(np.float32) - (sn * y).astype(np.float32)).astype(np.float32)
                 + (vx * tau).astype(np.float32)).astype(np.float32)
    out[:, 1] = (((sn * x).astype(np.float32) + (cn * y).astype(np.float32)).astype(np.float32)
                 + (vy * tau).astype(np.float32)).astype(np.float32)
    return out
