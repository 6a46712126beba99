"""Deterministic synthetic raw scans (arrays of the reference's 8-byte ``node_hq`` records).

Product-side input generator for ``bench.py`` and the tests (the CPU oracle is NOT used
to make inputs).  Every scan is a pure function of ``(seed, scan_index, n, options)`` so a
rank that owns scans ``[lo, hi)`` of a batch generates exactly the same bytes the
single-GPU run sees for those indices.

Shapes follow SURVEY.md §8(d): near-uniform Q14 angles (optional jitter / rotation to
exercise unsorted input), ~10 % invalid samples (``dist_mm_q2 == 0``) in runs so the
head / tail / fill branches of ``ascendScanData`` all fire, and either a smooth ring
``r0 + a*sin(k*theta + phi)`` with ``r0`` in [1, 30] m or uniformly random ranges.

The reference's own fake backend (``DummyLidarDriver::grab_scan_data``,
src/lidar_driver_wrapper.cpp:441-471, config 1) is mirrored in C++ under ``host/``.
"""
from __future__ import annotations

import numpy as np

from .abi import NODE_DTYPE


def make_scan(seed: int, scan_index: int, n: int, *, invalid_p: float = 0.10,
              run_len: float = 8.0, jitter: int = 0, rotate: bool = False,
              kind: str = "ring", r0_range=(1.0, 30.0), noise_m: float = 0.0,
              new_protocol: bool = False) -> np.ndarray:
    """One synthetic scan of ``n`` samples."""
    rng = np.random.default_rng(np.random.SeedSequence([int(seed), int(scan_index)]))
    nodes = np.zeros(n, NODE_DTYPE)
    if n == 0:
        return nodes
    i = np.arange(n, dtype=np.int64)
    q = (i * 65536) // n
    if jitter > 0:
        q = np.clip(q + rng.integers(-jitter, jitter + 1, n), 0, 65535)
    theta = q.astype(np.float64) * (2.0 * np.pi / 65536.0)
    if kind == "ring":
        r0 = rng.uniform(*r0_range)
        a = rng.uniform(0.0, 0.3) * r0
        k = int(rng.integers(1, 9))
        phi = rng.uniform(0.0, 2.0 * np.pi)
        r = r0 + a * np.sin(k * theta + phi)
        if noise_m > 0.0:
            r = r + rng.normal(0.0, noise_m, n)
        dist = np.maximum(r * 4000.0, 1.0).astype(np.uint32)
    elif kind == "uniform":
        dist = rng.integers(600, 160001, n).astype(np.uint32)
    else:
        raise ValueError(kind)
    if invalid_p > 0.0:
        nruns = int(round(invalid_p * n / run_len))
        if nruns > 0:
            starts = rng.integers(0, n, nruns)
            lens = rng.geometric(1.0 / run_len, nruns)
            diff = np.zeros(n + 1, np.int32)
            np.add.at(diff, starts, 1)
            np.add.at(diff, np.minimum(starts + lens, n), -1)
            dist[np.cumsum(diff[:-1]) > 0] = 0
    if new_protocol:
        quality = rng.integers(0, 256, n)
    else:
        quality = rng.integers(0, 64, n) << 2
    nodes["angle_z_q14"] = q.astype(np.uint16)
    nodes["dist_mm_q2"] = dist
    nodes["quality"] = quality.astype(np.uint8)
    nodes["flag"][0] = 1
    if rotate:
        shift = int(rng.integers(0, n))
        nodes = np.roll(nodes, shift)
    return nodes


def make_batch(seed: int, B: int, n: int, *, first_scan: int = 0, **kw) -> np.ndarray:
    """``B`` scans ``[first_scan, first_scan + B)`` of ``n`` samples, shape ``(B, n)``."""
    out = np.zeros((B, n), NODE_DTYPE)
    for s in range(B):
        out[s] = make_scan(seed, first_scan + s, n, **kw)
    return out
