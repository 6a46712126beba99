"""GPU parity at the BENCH scale and on the other data regimes.

``bench.py`` measures BASELINE config 3 — seed 2026, 4096 scans x 32 000 samples through
``rplgpu_cloud_arena_dev`` — and reports noisy / uniform variants of the same shape.  These
tests run exactly those batches through the same entry point and compare with the CPU oracle:
  * every k-th scan in full (number of cells, (iy, ix) order, centroids <= 1e-6 m, mean
    intensity bit-exact),
  * the whole batch through its cell count (the oracle on all host cores),
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


def _run_arena(gpu, batch, p, cap_per_scan):
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
    # the optional cell-key output (include/rplgpu.h): one word per output point, same index
    d_keys = torch.zeros(cap, dtype=torch.int32, device=dev)
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
            d_keys[:total].cpu().numpy().view(np.uint32))


def _check_batch(gpu, oracle, batch, p, every, cap_per_scan):
    B, n = batch.shape
    arena, start, npts, st, total, keys = _run_arena(gpu, batch, p, cap_per_scan)
    assert int(st.max()) == 0, "status_bits"
    assert total == int(npts.sum())
    # the reservations tile [0, total) without gaps or overlaps
    nz = np.nonzero(npts)[0]
    order = nz[np.argsort(start[nz], kind="stable")]
    assert np.array_equal(np.cumsum(npts[order]) - npts[order], start[order])
    # whole batch: the oracle's cell count (all host cores)
    op = oracle_lib.copy_params(p)
    lens = np.full(B, n, np.uint32)
    nodes = np.ascontiguousarray(batch)
    want_total = int(oracle.lib.orc_batch_cloud(nodes.ctypes.data, n, lens.ctypes.data, B,
                                                C.byref(op), os.cpu_count() or 1))
    assert total == want_total
    worst = 0.0
    for b in range(0, B, every):
        want, wcells, wcounts = oracle.cloud_pipeline(batch[b], op)
        got = arena[start[b]: start[b] + npts[b]]
        assert len(got) == len(want), b
        if not len(want):
            continue
        err = np.max(np.abs(got[:, :2].astype(np.float64) - want[:, :2]))
        worst = max(worst, float(err))
        assert err <= XYZ_TOL, (b, err)
        assert np.all(got[:, 2] == 0.0)
        assert got[:, 3].tobytes() == want[:, 3].tobytes(), b  # mean intensity: bit-exact
        # cell indices: bit-exact.  The kernel's optional cell-key output carries the (iy, ix) every
        # output point was reduced under; it must be the oracle's cell list, in the oracle's order
        k = keys[start[b]: start[b] + npts[b]]
        wkey = ((wcells[:, 1].astype(np.int64) + 32768) << 16 | (wcells[:, 0].astype(np.int64) + 32768))
        assert np.array_equal(k.astype(np.int64), wkey), b
        assert np.all(np.diff(k.astype(np.int64)) > 0), b  # strictly ascending (iy, ix)
    return total, worst


def test_bench_batch_config3_voxel_matches_oracle(gpu, oracle):
    """The bench batch itself: seed 2026, 4096 x 32 000, the bench parameters."""
    B, n = 4096, 32000
    batch = synth.make_batch(2026, B, n)
    p = Params.defaults(clip_enable=1, q_min=0, range_min=0.15, range_max=40.0, voxel_enable=1,
                        voxel_leaf=0.05)
    total, worst = _check_batch(gpu, oracle, batch, p, every=64, cap_per_scan=8192)
    assert total > 9_000_000  # ~2.7 k cells per scan
    # the same batch with the quality filter of BASELINE config 3 switched on
    pq = Params.defaults(clip_enable=1, q_min=48, range_min=0.15, range_max=40.0, voxel_enable=1,
                         voxel_leaf=0.05)
    total_q, _ = _check_batch(gpu, oracle, batch[:512], pq, every=32, cap_per_scan=8192)
    assert 0 < total_q


@pytest.mark.parametrize("regime", ["ring_noise_1cm", "uniform"])
def test_other_regimes_at_full_scan_size(gpu, oracle, regime):
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
    _check_batch(gpu, oracle, batch, p, every=8, cap_per_scan=n)


def test_c5_shape_ror_voxel_batch_matches_oracle(gpu, oracle):
    """Config 5 at batch scale: 8 sensors x 8 frames of 32 000 noisy samples, E5 + E4."""
    B, n = 64, 32000
    batch = synth.make_batch(2031, B, n, noise_m=0.01)
    p = Params.defaults(clip_enable=1, range_min=0.15, range_max=40.0, voxel_enable=1,
                        voxel_leaf=0.05, ror_enable=1, ror_radius=0.10, ror_min_neighbors=2)
    _check_batch(gpu, oracle, batch, p, every=16, cap_per_scan=n)
