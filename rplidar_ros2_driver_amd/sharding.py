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


# ---------------------------------------------------------------------------------------------
# The exchange behind the C ABI (include/rplgpu_comm.h): layout twins + the chunked, overlapped
# driver bench.py uses.  The twins restate, in torch, what k_pack_meta / k_unpack_gathered do, so
# that the N > 1 layout is testable on CPU (gloo, world_size 2) and the kernels are checked
# against them on the GPU.
# ---------------------------------------------------------------------------------------------
def meta_words(max_scans: int) -> int:
    return 4 + 3 * max_scans


def pack_cloud_meta(cursor: int, scan_start, n_points, slot_points: int, max_scans: int) -> torch.Tensor:
    """Twin of ``rplgpu_pack_cloud_meta_dev``: int64 inputs -> (meta_words,) int32 tensor holding
    the u32 words [count lo, count hi, B, flags, B x {start lo, start hi, n_points}]."""
    B = int(len(n_points))
    assert B <= max_scans
    m = torch.zeros(meta_words(max_scans), dtype=torch.int64)
    k = min(int(cursor), int(slot_points))
    m[0], m[1], m[2], m[3] = k & 0xFFFFFFFF, k >> 32, B, int(int(cursor) > int(slot_points))
    for s in range(B):
        st, n = int(scan_start[s]), int(n_points[s])
        if st >= slot_points:
            st, n = 0, 0
        elif st + n > slot_points:
            n = int(slot_points) - st
        m[4 + 3 * s], m[5 + 3 * s], m[6 + 3 * s] = st & 0xFFFFFFFF, st >> 32, n
    return m.to(torch.int32) if int(m.max()) < 2**31 else (m - (m >= 2**31) * 2**32).to(torch.int32)


def unpack_gathered(points_all: torch.Tensor, meta_all: torch.Tensor, slot_points: int,
                    world: int, max_scans: int):
    """Twin of ``rplgpu_unpack_gathered_dev``: (world*slot_points, 4) float32 slots and
    (world, meta_words) int32 meta -> (packed (total, 4), scan_start_all (world, max_scans) int64,
    n_points_all (world, max_scans) int64, status (world,))."""
    meta = meta_all.view(world, -1).to(torch.int64) & 0xFFFFFFFF
    counts = torch.clamp(meta[:, 0] | (meta[:, 1] << 32), max=slot_points)
    offs = torch.cumsum(counts, 0) - counts
    packed = torch.cat([points_all.view(world, slot_points, 4)[r, : int(counts[r])] for r in range(world)]) \
        if world else points_all.new_zeros(0, 4)
    starts = torch.zeros(world, max_scans, dtype=torch.int64)
    npts = torch.zeros(world, max_scans, dtype=torch.int64)
    for r in range(world):
        B = min(int(meta[r, 2]), max_scans)
        for s in range(B):
            n = int(meta[r, 6 + 3 * s])
            npts[r, s] = n
            starts[r, s] = (int(meta[r, 4 + 3 * s]) | (int(meta[r, 5 + 3 * s]) << 32)) + int(offs[r]) if n else 0
    status = (meta[:, 3] & 1) * 8  # RPLGPU_SCAN_OUT_TRUNCATED
    return packed, starts, npts, status


def gather_slots_to_root(slot: torch.Tensor, meta: torch.Tensor, root: int, group=None):
    """Twin of ``rplgpu_gather_clouds_dev`` over ``torch.distributed`` (gloo on CPU, RCCL on GPU):
    every rank's fixed-size slot and META block meet on ``root`` in the layout of the all-gather
    (rank-major); the other ranks only send and get ``(None, None)``.  One slot per link, into the
    root — what BASELINE config 5 needs (the fused message is built on one GPU)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if rank == root:
        slots = [torch.empty_like(slot) for _ in range(world)]
        metas = [torch.empty_like(meta) for _ in range(world)]
        dist.gather(slot, slots, dst=root, group=group)
        dist.gather(meta, metas, dst=root, group=group)
        return torch.stack(slots), torch.stack(metas)
    dist.gather(slot, None, dst=root, group=group)
    dist.gather(meta, None, dst=root, group=group)
    return None, None


# Launch time of the voxel kernel against the scans of ONE launch on one MI355X (256 compute units, 32 000-sample
# ring scans; profiles/r05/voxel_instruction_mix_r05.txt section 5), relative to the 4096-scan launch: a launch of
# few scans pays its fixed cost (launch, prologue, the tail of a dynamically drawn queue) and, below one scan per
# compute unit, leaves units idle.  What a CHUNK of a step costs — the reason more chunks are not always better.
_LAUNCH_REL = ((0, 0.06), (128, 0.12), (256, 0.1376), (512, 0.2180), (1024, 0.3131), (2048, 0.5365), (4096, 1.0),
               (8192, 1.9333), (16384, 3.8173))
_EXCHANGE_FIXED_MS = 0.015  # per chunk: the collective's launch + the unpack kernel (one-rank runs, 1 / 4 / 8 chunks)


def launch_rel(scans: float) -> float:
    """Interpolated `_LAUNCH_REL` (linear between the measured points, the last slope beyond them)."""
    pts = _LAUNCH_REL
    if scans >= pts[-1][0]:
        (x0, y0), (x1, y1) = pts[-2], pts[-1]
        return y1 + (scans - x1) * (y1 - y0) / (x1 - x0)
    for (x0, y0), (x1, y1) in zip(pts, pts[1:]):
        if scans <= x1:
            return y0 + (max(scans, x0) - x0) * (y1 - y0) / (x1 - x0)
    return pts[-1][1]


def predicted_step_ms(world: int, compute_ms_one_gpu: float, cloud_bytes_total: float, chunks: int,
                      link_gbs: float = 76.8, scans_total: int = 4096) -> float:
    """DESIGN.md section 7's model of a step on `world` GPUs (strong scaling of one batch of `scans_total`
    scans): every rank computes 1/world of the batch in `chunks` launches — a launch priced with the measured
    launch-time curve, scaled to `compute_ms_one_gpu` (the whole batch in one launch on one GPU) — a piece's
    exchange starts when the piece is computed and pieces follow one another on the links.  All-gather: a link
    carries one rank's share of a piece (the links of a rank work in parallel).  Gather to root is priced THE
    SAME (the model used to take a `gather_root` flag and ignore it): the step is the slowest rank's,
    that is the root's — its world - 1 inbound slots arrive on world - 1 different links, one slot per link
    like the all-gather's, and it runs the same unpack; what the gather saves (no inbound traffic, no unpack on
    the other ranks, 1/world of the HBM writes) is not on the step's critical path.  Whether the grouped
    send / receive costs more or less per piece than ncclAllGather is not known: no run with two ranks has
    ever been made.  Every piece also pays the exchange's fixed cost.  Returns milliseconds."""
    if world <= 1:
        return compute_ms_one_gpu
    comp = compute_ms_one_gpu * launch_rel(scans_total / world / chunks) / launch_rel(scans_total)
    exch = cloud_bytes_total / world / chunks / (link_gbs * 1e6) + _EXCHANGE_FIXED_MS  # one piece on one link, ms
    t_c = t_x = 0.0
    for _ in range(chunks):
        t_c += comp
        t_x = max(t_x, t_c) + exch
    return t_x


def best_chunks(world: int, compute_ms_one_gpu: float, cloud_bytes_total: float, scans_total: int,
                candidates=(1, 2, 4, 8)) -> int:
    """The chunk count the model prefers (ties: fewer chunks).  One rank: one chunk — there is nothing to hide."""
    if world <= 1:
        return 1
    best, best_ms = 1, float("inf")
    for c in candidates:
        if c > max(scans_total // world, 1):
            break
        ms = predicted_step_ms(world, compute_ms_one_gpu, cloud_bytes_total, c, scans_total=scans_total)
        if ms < best_ms * 0.995:
            best, best_ms = c, ms
    return best


class CloudExchange:
    """bench.py's N > 1 step through the library's own exchange: the rank's block of scans is cut
    into `chunks`; chunk c is voxelised as 12-byte points (x, y, intensity: z is 0 for every point
    of this path and does not travel) by the voxel kernel itself STRAIGHT INTO this
    rank's slot of chunk c's receive buffer (RCCL's in-place all-gather then moves no local
    bytes) on the handle's main stream, and all-gathered on the exchange stream while chunk c + 1
    is being voxelised.  Every chunk has its own receive buffer, so nothing is reused inside a step and the
    step needs no fence but the last.  Slot sizes are fixed after one calibration pass (max cells
    per chunk over all ranks + 5 % head room; a cloud that outgrows its slot is cut and flagged,
    never overrun), so a timed step has no host synchronisation: counts travel in the META blocks
    on the device."""

    def __init__(self, gpu, dist_mod, dev, world, rank, B, n, out_stride, chunks, root=None):
        # root: None = all-gather (every rank ends up with the whole cloud, BASELINE config 4);
        # an int = gather to that rank only (rplgpu_gather_clouds_dev, config 5's fused message)
        self.root = root
        self.gpu, self.dist, self.dev, self.world, self.rank = gpu, dist_mod, dev, world, rank
        self.B, self.n, self.out_stride = B, n, out_stride
        self.chunks = max(1, min(chunks, max(B, 1)))
        self.Bc = (B + self.chunks - 1) // self.chunks
        self.mw = meta_words(self.Bc)
        # communicator: rank 0's unique id travels through torch.distributed.  Should the library's
        # own communicator not come up on some rank (RCCL not loadable, init error), EVERY rank
        # falls back to torch.distributed's all-gather for the same buffers, and `backend` says so
        # (RPL_EXCHANGE=torch forces that path).
        import os
        self.backend = "rccl behind the C ABI (rplgpu_comm.h)"
        ok = 1
        if os.environ.get("RPL_EXCHANGE") == "torch":
            ok, self.backend = 0, "torch.distributed (forced by RPL_EXCHANGE=torch)"
        else:
            uid = torch.zeros(128, dtype=torch.uint8, device=dev)
            try:
                if rank == 0:
                    uid.copy_(torch.from_numpy(gpu.comm_unique_id()))
            except Exception as e:  # (the broadcast below still has to happen on every rank)
                ok, self.backend = 0, f"torch.distributed (rplgpu_comm_unique_id failed: {e})"
            self.dist.broadcast(uid, src=0)
            # rplgpu_comm_init is a collective: agree on "the id is good" BEFORE any rank enters it
            # (rank 0 failing to make the id must not leave the others waiting inside RCCL)
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN)
            if ok and int(flag.item()) == 0:
                self.backend = "torch.distributed (rank 0 could not make a communicator id)"
            ok = int(flag.item())
            if ok:
                try:
                    gpu.comm_init(rank, world, uid.cpu().numpy())
                except Exception as e:
                    ok, self.backend = 0, f"torch.distributed (rplgpu_comm_init failed: {e})"
                flag = torch.tensor([ok], dtype=torch.int32, device=dev)
                self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN)
                if ok and int(flag.item()) == 0:
                    self.backend = "torch.distributed (a peer's communicator did not come up)"
                    gpu.comm_destroy()
                ok = int(flag.item())
        self.native = bool(ok)
        # what RCCL itself says about the communicator (bench.py refuses a run whose communicator
        # does not span the ranks it was asked for)
        self.comm_ranks = gpu.comm_size()[0] if self.native else None
        self.cursor = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(self.chunks)]
        self.start = [torch.zeros(self.Bc, dtype=torch.int64, device=dev) for _ in range(self.chunks)]
        self.npts = [torch.zeros(self.Bc, dtype=torch.int32, device=dev) for _ in range(self.chunks)]
        self.stat = [torch.zeros(self.Bc, dtype=torch.int32, device=dev) for _ in range(self.chunks)]
        self.slot = None
        self.recv_pts = None  # per chunk: every rank's compact slot (12-byte points: x, y, intensity)
        self.recv_meta = torch.zeros(self.chunks, world, self.mw, dtype=torch.int32, device=dev)

    def _chunk(self, c):
        lo = c * self.Bc
        return lo, min(self.Bc, self.B - lo)

    def _calibrate(self, d_nodes, d_len, params):
        cap = self.Bc * self.out_stride
        tmp = torch.empty(cap, 4, dtype=torch.float32, device=self.dev)
        worst = 0
        for c in range(self.chunks):
            lo, nb = self._chunk(c)
            if nb <= 0:
                continue
            self.gpu.cloud_arena_dev(d_nodes.data_ptr() + lo * self.n * 8, self.n, d_len.data_ptr() + lo * 4,
                                     nb, params, tmp.data_ptr(), cap, self.cursor[0].data_ptr(),
                                     self.start[0].data_ptr(), self.npts[0].data_ptr(),
                                     self.stat[0].data_ptr())
            self.gpu.synchronize()
            worst = max(worst, int(self.cursor[0].item()))
        del tmp
        t = torch.tensor([worst], dtype=torch.int64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        # head room 5 % + 256 points: the slot travels whole (an all-gather moves equal pieces)
        self.slot = min(cap, int(int(t.item()) * 1.05) + 256)
        self.recv_pts = torch.empty(self.chunks, self.world, self.slot, 3, dtype=torch.float32,
                                    device=self.dev)

    def step(self, d_nodes, d_len, params):
        if self.slot is None:
            self._calibrate(d_nodes, d_len, params)
        g, r = self.gpu, self.rank
        for c in range(self.chunks):
            lo, nb = self._chunk(c)
            if nb <= 0:
                continue
            mine = self.recv_pts[c, r]      # this rank's slot of chunk c (in place in the receive buffer)
            meta = self.recv_meta[c, r]
            # (round 4: the voxel kernel writes the 12-byte points straight into the slot — the
            # 16-byte arena and the compaction pass in front of the all-gather are gone)
            g.cloud_arena_xyi_dev(d_nodes.data_ptr() + lo * self.n * 8, self.n, d_len.data_ptr() + lo * 4,
                                  nb, params, mine.data_ptr(), self.slot, self.cursor[c].data_ptr(),
                                  self.start[c].data_ptr(), self.npts[c].data_ptr(),
                                  self.stat[c].data_ptr())
            g.pack_cloud_meta_dev(self.cursor[c].data_ptr(), self.start[c].data_ptr(),
                                  self.npts[c].data_ptr(), nb, self.slot, self.Bc, meta.data_ptr())
            self._gather(c, mine, meta)
        if self.native:
            g.comm_fence(0)

    def _gather(self, c, mine, meta):
        if self.native and self.root is not None:
            self.gpu.gather_clouds_dev(self.root, mine.data_ptr(), self.slot, 3, meta.data_ptr(), self.mw,
                                       self.recv_pts[c].data_ptr(), self.recv_meta[c].data_ptr())
        elif self.native:
            self.gpu.allgather_clouds_xyi_dev(mine.data_ptr(), self.slot, meta.data_ptr(), self.mw,
                                              self.recv_pts[c].data_ptr(), self.recv_meta[c].data_ptr())
        elif self.root is not None:
            self.gpu.synchronize()
            slots, metas = gather_slots_to_root(mine.reshape(-1).clone(), meta.reshape(-1).clone(), self.root)
            if slots is not None:
                self.recv_pts[c].view(self.world, -1).copy_(slots)
                self.recv_meta[c].view(self.world, -1).copy_(metas)
        else:  # fallback: torch.distributed on torch's current stream — the handle's stream (which
            # the kernels above were queued on) must be drained first; a send copy to be safe
            self.gpu.synchronize()
            self.dist.all_gather_into_tensor(self.recv_pts[c].view(-1), mine.reshape(-1).clone())
            self.dist.all_gather_into_tensor(self.recv_meta[c].view(-1), meta.reshape(-1).clone())

    def exchange_only(self):
        r = self.rank
        for c in range(self.chunks):
            self._gather(c, self.recv_pts[c, r], self.recv_meta[c, r])
        if self.native:
            self.gpu.comm_fence(0)

    def last_bytes(self):
        """bytes this rank RECEIVES from its peers per step"""
        if self.root is not None and self.rank != self.root:
            return 0
        return int(self.chunks * (self.world - 1) * (self.slot or 0) * 12)

    def unpack(self, c, d_packed, d_total, d_start_all, d_np_all, d_status=0):
        """chunk c's gathered slots -> one contiguous cloud of 16-byte points + per-scan tables"""
        self.gpu.unpack_gathered_xyi_dev(self.recv_pts[c].data_ptr(), self.slot,
                                         self.recv_meta[c].data_ptr(), self.mw, self.world, self.Bc,
                                         d_packed, d_total, d_start_all, d_np_all, d_status)

    def close(self):
        if self.native:
            self.gpu.comm_destroy()
