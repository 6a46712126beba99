"""Scan-index sharding and the all-gather of filtered clouds.

Every scan is independent, so a batch shards by contiguous scan index with no exchange
needed to *compute*; one exchange *assembles* the result: a variable-length all-gather
of the per-rank packed clouds.  With ``torch.distributed`` backend ``nccl`` this is RCCL
over xGMI (fully connected, point-to-point links): counts first (3 x int64 per rank), then one
padded ``all_gather_into_tensor`` — the per-scan table riding behind the points — so each link
carries 1/G of the payload exactly once; no ring all-reduce anywhere.  The same code runs on ``gloo`` with CPU tensors, which is
how the N>1 logic is tested without GPUs.
"""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous block ``[lo, hi)`` of ``total`` scans owned by ``rank`` (sizes differ by
    at most one; earlier ranks take the remainder)."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, rem = divmod(total, world_size)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def allgather_clouds(packed: torch.Tensor, n_points, scan_counts: torch.Tensor,
                     group=None, scan_starts: torch.Tensor | None = None):
    """All-gather variable-length clouds.

    packed       (cap, 4) float32 — this rank's contiguous cloud, first ``n_points`` rows valid;
                 rows past ``n_points`` are scratch (the per-scan table may be written there)
    n_points     int, or a 1-element integer tensor on ``packed``'s device (the arena cursor; it
                 is clamped to the capacity, as the kernel clamps what it writes)
    scan_counts  (B_local,) int32/int64 — points per local scan (so receivers can split)
    scan_starts  optional (B_local,) int64 — first row of every local scan when the cloud is an
                 *arena* (``rplgpu_cloud_arena_dev``: scans in completion order); without it the
                 scans are taken to lie back to back in scan order (``rplgpu_pack_clouds_dev``)
    Returns ``(clouds, counts)`` or, with ``scan_starts``, ``(clouds, counts, starts)``:
    per-rank lists of ``(n_r, 4)`` tensors (views into one receive buffer) and of per-scan
    counts / starts, in rank == scan order.
    """
    world = dist.get_world_size(group)
    dev = packed.device
    if torch.is_tensor(n_points):  # e.g. the arena cursor, still on the device: no host round trip
        meta = torch.cat([n_points.reshape(1).to(torch.int64).clamp(max=int(packed.shape[0])),
                          torch.tensor([int(scan_counts.numel()), int(packed.shape[0])],
                                       dtype=torch.int64, device=dev)])
    else:
        meta = torch.tensor([int(n_points), int(scan_counts.numel()), int(packed.shape[0])],
                            dtype=torch.int64, device=dev)
    # outputs are allocated flat (concatenation along dim 0): the one layout both the
    # nccl (RCCL) and gloo implementations of all_gather_into_tensor accept
    metas = torch.empty(world * 3, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(metas, meta, group=group)
    metas_h = metas.view(world, 3).cpu()
    max_pts = int(metas_h[:, 0].max())
    max_scans = int(metas_h[:, 1].max())

    # per-scan counts (and arena starts) as one small int64 table
    cols = 2 if scan_starts is not None else 1
    table = torch.zeros(cols, max_scans, dtype=torch.int64, device=dev)
    table[0, : scan_counts.numel()] = scan_counts.to(torch.int64)
    if scan_starts is not None:
        table[1, : scan_starts.numel()] = scan_starts.to(torch.int64)
    table_rows = (cols * max_scans * 8 + 15) // 16  # cloud rows (16 B) the table occupies

    # When every rank's buffer has room behind max_pts (an arena always has: it is sized for the
    # worst case), the table rides in those rows and ONE collective moves everything; the rows
    # past n_points are scratch by contract.  All ranks take the same branch: the capacities
    # travelled with the counts.
    in_band = (packed.dtype == torch.float32 and packed.is_contiguous()
               and int(metas_h[:, 2].min()) >= max_pts + table_rows)
    allgather_clouds.last_in_band = bool(in_band)  # (for the tests)
    if in_band:
        rows = max_pts + table_rows
        if table_rows:
            packed[max_pts:rows].view(torch.int64).view(-1)[: cols * max_scans] = table.view(-1)
        send = packed[:rows].view(-1)
        recv = torch.empty(world * rows * 4, dtype=packed.dtype, device=dev)
        dist.all_gather_into_tensor(recv, send, group=group)
        recv = recv.view(world, rows, 4)
        if table_rows:
            cnt_recv = recv[:, max_pts:].reshape(world, -1).view(torch.int64)[:, : cols * max_scans]
            cnt_recv = cnt_recv.reshape(world, cols, max_scans)
        else:
            cnt_recv = torch.zeros(world, cols, 0, dtype=torch.int64, device=dev)
    else:
        send = packed[:max_pts] if packed.shape[0] >= max_pts else torch.cat(
            [packed, packed.new_zeros(max_pts - packed.shape[0], 4)])
        send = send.contiguous().view(-1)
        recv = torch.empty(world * max_pts * 4, dtype=packed.dtype, device=dev)
        dist.all_gather_into_tensor(recv, send, group=group)
        recv = recv.view(world, max_pts, 4)
        cnt_recv = torch.empty(world * cols * max_scans, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(cnt_recv, table.view(-1), group=group)
        cnt_recv = cnt_recv.view(world, cols, max_scans)

    clouds: List[torch.Tensor] = []
    counts: List[torch.Tensor] = []
    starts: List[torch.Tensor] = []
    for r in range(world):
        clouds.append(recv[r, : int(metas_h[r, 0])])
        counts.append(cnt_recv[r, 0, : int(metas_h[r, 1])])
        if scan_starts is not None:
            starts.append(cnt_recv[r, 1, : int(metas_h[r, 1])])
    if scan_starts is not None:
        return clouds, counts, starts
    return clouds, counts


def split_by_scan(cloud: torch.Tensor, counts: torch.Tensor,
                  starts: torch.Tensor | None = None) -> List[torch.Tensor]:
    """Split one rank's contiguous cloud back into per-scan clouds (``starts``: arena layout)."""
    sizes = [int(c) for c in counts.cpu().tolist()]
    if starts is None:
        return list(torch.split(cloud[: sum(sizes)], sizes)) if sizes else []
    return [cloud[int(a): int(a) + n] for a, n in zip(starts.cpu().tolist(), sizes)]
