"""N > 1 path on CPU: world_size-2 gloo run of the scan-index sharding and the
variable-length all-gather of voxelised clouds (the same code path RCCL runs on GPUs).
The per-rank clouds come from the oracle here (no GPU in this container); what is under
test is the sharding arithmetic and the exchange."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

from rplidar_ros2_driver_amd import synth  # noqa: E402
from rplidar_ros2_driver_amd.sharding import allgather_clouds, shard_range, split_by_scan  # noqa: E402


def test_shard_range_covers_everything():
    for total in (0, 1, 7, 8, 4096, 4099):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _local_clouds(lo, hi, n):
    from tests import oracle_lib
    orc = oracle_lib.load_oracle()
    p = oracle_lib.params(clip_enable=1, range_max=40.0, voxel_enable=1)
    clouds = []
    for s in range(lo, hi):
        out, _, _ = orc.cloud_pipeline(synth.make_scan(77, s, n), p)
        clouds.append(out)
    return clouds


def _worker(rank, world, port, total, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(total, world, rank)
        clouds = _local_clouds(lo, hi, n)
        counts = torch.tensor([len(c) for c in clouds], dtype=torch.int32)
        packed = torch.from_numpy(np.concatenate(clouds + [np.zeros((0, 4), np.float32)]))
        cap = torch.zeros(int(counts.sum()) + 5, 4)  # capacity > valid rows, like the GPU buffer
        cap[: len(packed)] = packed
        got_clouds, got_counts = allgather_clouds(cap, int(counts.sum()), counts)
        # every rank must now hold every scan's cloud, in scan order
        full = []
        for r in range(world):
            full += split_by_scan(got_clouds[r], got_counts[r])
        ref = _local_clouds(0, total, n)
        ok = len(full) == total and all(
            f.numpy().tobytes() == w.tobytes() for f, w in zip(full, ref))
        # arena layout (rplgpu_cloud_arena_dev): the local scans lie in the cloud in completion
        # order, here reversed, and their starts travel with the counts
        order = list(range(len(clouds)))[::-1]
        starts = torch.zeros(len(clouds), dtype=torch.int64)
        arena = torch.zeros(int(counts.sum()) + 3, 4)
        at = 0
        for s_ in order:
            starts[s_] = at
            arena[at: at + len(clouds[s_])] = torch.from_numpy(clouds[s_])
            at += len(clouds[s_])
        a_clouds, a_counts, a_starts = allgather_clouds(arena, at, counts, scan_starts=starts)
        full2 = []
        for r in range(world):
            full2 += split_by_scan(a_clouds[r], a_counts[r], a_starts[r])
        ok = ok and len(full2) == total and all(
            f.numpy().tobytes() == w.tobytes() for f, w in zip(full2, ref))
        # a roomy arena (the real one is sized for the worst case): the per-scan table rides
        # behind the points and one collective moves everything
        roomy = torch.zeros(at + 4096, 4)
        roomy[:at] = arena[:at]
        ok = ok and allgather_clouds.last_in_band is False  # the tight arena above: two collectives
        b_clouds, b_counts, b_starts = allgather_clouds(roomy, at, counts, scan_starts=starts)
        ok = ok and allgather_clouds.last_in_band is True
        full3 = []
        for r in range(world):
            full3 += split_by_scan(b_clouds[r], b_counts[r], b_starts[r])
        ok = ok and roomy[:at].numpy().tobytes() == arena[:at].numpy().tobytes()  # points intact
        ok = ok and len(full3) == total and all(
            f.numpy().tobytes() == w.tobytes() for f, w in zip(full3, ref))
        c_clouds, c_counts = allgather_clouds(roomy, torch.tensor([at]), counts)  # device-side count
        ok = ok and all(torch.equal(c_counts[r], b_counts[r]) for r in range(world))
        ok = ok and all(c_clouds[r].shape == b_clouds[r].shape for r in range(world))
        q.put((rank, bool(ok), len(full)))
    finally:
        dist.destroy_process_group()


def test_step_model_and_chunk_choice():
    """sharding.predicted_step_ms / best_chunks: the measured launch-time curve makes a
    chunk a launch with a fixed cost, so more chunks are not always better; one rank never chunks."""
    from rplidar_ros2_driver_amd.sharding import best_chunks, launch_rel, predicted_step_ms
    assert launch_rel(4096) == 1.0 and launch_rel(8192) > 1.9 and launch_rel(16384) > 3.8
    xs = [0, 64, 128, 256, 300, 512, 1000, 2048, 4096, 9000, 20000, 40000]
    ys = [launch_rel(x) for x in xs]
    assert all(b > a for a, b in zip(ys, ys[1:]))               # monotone
    assert launch_rel(256) / launch_rel(4096) > 256 / 4096 * 2  # a small launch is far from proportional
    one = 0.449
    assert predicted_step_ms(1, one, 131e6, 4) == one and best_chunks(1, one, 131e6, 4096) == 1
    for world in (2, 4, 8):
        ms = {c: predicted_step_ms(world, one, 131e6, c, scans_total=4096) for c in (1, 2, 4, 8)}
        best = best_chunks(world, one, 131e6, 4096)
        assert ms[best] <= min(ms.values()) * 1.006
        assert ms[2] < ms[1]                       # hiding half the compute under the exchange pays ...
        assert ms[8] > min(ms.values())            # ... eight launches per step do not
        # the exchange is the step from two GPUs on: never faster than the slot on one link
        assert min(ms.values()) > 131e6 / world / 76.8e6
    assert best_chunks(8, one, 131e6, 4096) == 2   # 512 scans per rank: a quarter of them is half a GPU
    assert best_chunks(4, one, 131e6, 8) <= 2      # never more chunks than scans per rank


def test_allgather_clouds_world2_gloo():
    world, total, n = 2, 5, 1500  # uneven shard: 3 + 2 scans
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res), res


# ---- BASELINE config 5 across ranks: one sensor per rank -> gather -> transform -> fused message
def _c5_worker(rank, world, port, per_rank, n, q):
    """The layout of include/rplgpu_comm.h end to end on CPU: every rank voxelises ITS sensors
    (oracle clouds stand in for the GPU kernels here), lays them out as rplgpu_cloud_arena_dev
    does (completion order != scan order), builds the META block, all-gathers fixed-size slots
    and META blocks (gloo), unpacks them into one cloud + per-scan tables, applies the per-sensor
    poses and serialises ONE PointCloud2 — which must equal, byte for byte, what a single process
    holding all sensors produces."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, str(ROOT / "oracle"))
        import cdr_oracle
        import fusion_oracle as fo
        from rplidar_ros2_driver_amd import sharding as sh
        from tests import oracle_lib
        orc = oracle_lib.load_oracle()
        p = oracle_lib.params(clip_enable=1, range_max=40.0, voxel_enable=1)
        S = world * per_rank
        poses = np.stack([fo.planar_pose(0.7 * s - 1.0, 0.35 * s, -0.2 * s, 0.05 * s) for s in range(S)])

        def cloud_of(sensor):
            return orc.cloud_pipeline(synth.make_scan(900 + sensor, 0, n, noise_m=0.01), p)[0]

        # this rank's arena: its sensors in REVERSED completion order, a slot with head room
        mine = [cloud_of(rank * per_rank + j) for j in range(per_rank)]
        counts = [len(c) for c in mine]
        slot = max(8000, sum(counts) + 100)
        t = torch.tensor([slot], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        slot = int(t.item())
        arena = torch.full((slot, 4), -7.0)
        starts, at = [0] * per_rank, 0
        for j in reversed(range(per_rank)):
            starts[j] = at
            arena[at: at + counts[j]] = torch.from_numpy(mine[j])
            at += counts[j]
        # the layout is the PRODUCT's: the C ABI's host entry points (include/rplgpu_comm.h, the
        # same rules the device kernels compile: csrc/rpl_comm_layout.hpp); the torch twins of
        # sharding.py are only cross-checked against them
        from rplidar_ros2_driver_amd import abi
        meta_np = abi.pack_cloud_meta_host(at, starts, counts, slot, per_rank)
        meta = torch.from_numpy(meta_np.view(np.int32).copy())
        ok = torch.equal(meta, sh.pack_cloud_meta(at, starts, counts, slot, per_rank))
        pts_all = torch.empty(world * slot * 4)
        dist.all_gather_into_tensor(pts_all, arena.view(-1))
        meta_all = torch.empty(world * sh.meta_words(per_rank), dtype=torch.int32)
        dist.all_gather_into_tensor(meta_all, meta)
        packed_np, st_np, npn_np, status_np = abi.unpack_gathered_host(
            pts_all.view(world, slot, 4).numpy(), slot, meta_all.numpy().view(np.uint32), world, per_rank)
        packed, st_all, np_all, status = (torch.from_numpy(packed_np.copy()), st_np.astype(np.int64),
                                          npn_np.astype(np.int64), status_np.astype(np.int64))
        t_packed, t_st, t_np, t_status = sh.unpack_gathered(pts_all.view(-1, 4), meta_all, slot, world,
                                                            per_rank)
        ok = ok and torch.equal(packed, t_packed) and np.array_equal(st_all, t_st.numpy()) \
            and np.array_equal(np_all, t_np.numpy()) and np.array_equal(status, t_status.numpy())
        ok = ok and int(status.sum()) == 0 and len(packed) == int(np_all.sum())
        # the compact exchange: 12-byte points (x, y, intensity) travel, z = 0 comes back
        slot12 = torch.from_numpy(abi.pack_cloud_xyi_host(arena.numpy(), at, slot))
        s12_all = torch.empty(world * slot * 3)
        dist.all_gather_into_tensor(s12_all, slot12.view(-1))
        packed12, st12, np12, status12 = abi.unpack_gathered_host(
            s12_all.view(world, slot, 3).numpy(), slot, meta_all.numpy().view(np.uint32), world, per_rank)
        ok = ok and packed12.tobytes() == packed_np.tobytes() and np.array_equal(st12, st_np) \
            and np.array_equal(np12, npn_np) and int(status12.sum()) == 0
        # gather to ONE rank (rplgpu_gather_clouds_dev's layout: the all-gather's, on the root only)
        root = world - 1
        g_slots, g_metas = sh.gather_slots_to_root(slot12.reshape(-1), meta, root)
        if rank == root:
            gp, gs, gn, gst = abi.unpack_gathered_host(
                g_slots.view(world, slot, 3).numpy(), slot, g_metas.numpy().view(np.uint32), world, per_rank)
            ok = ok and gp.tobytes() == packed_np.tobytes() and np.array_equal(gs, st_np) \
                and np.array_equal(gn, npn_np) and int(gst.sum()) == 0
        else:
            ok = ok and g_slots is None and g_metas is None
        # per-sensor transform into the common frame, then one serialised PointCloud2
        fused = packed.numpy().copy()
        for r in range(world):
            for j in range(per_rank):
                a, k = int(st_all[r, j]), int(np_all[r, j])
                fused[a: a + k] = fo.transform_cloud(fused[a: a + k], poses[r * per_rank + j])
        msg = cdr_oracle.cloud_msg("base_link", 12, 34, fused)
        # single process: all sensors, arena order = rank-major, each rank's reversed order
        ref_parts = []
        for r in range(world):
            for j in reversed(range(per_rank)):
                s = r * per_rank + j
                ref_parts.append(fo.transform_cloud(cloud_of(s), poses[s]))
        ref_msg = cdr_oracle.cloud_msg("base_link", 12, 34, np.concatenate(ref_parts))
        ok = ok and msg == ref_msg
        # a slot that is too small truncates and flags instead of overrunning
        small = max(1, at // 2)
        meta_s = abi.pack_cloud_meta_host(at, starts, counts, small, per_rank)
        ok = ok and int(meta_s[3]) == 1 and int(meta_s[0]) == small
        ok = ok and torch.equal(torch.from_numpy(meta_s.view(np.int32).copy()),
                                sh.pack_cloud_meta(at, starts, counts, small, per_rank))
        cut, _, np_cut, st_cut = abi.unpack_gathered_host(
            np.stack([abi.pack_cloud_xyi_host(arena.numpy(), at, small)] * world), small,
            np.stack([meta_s] * world), world, per_rank)
        ok = ok and len(cut) == world * small and int(np_cut.sum()) == world * small \
            and all(int(x) == 8 for x in st_cut)  # RPLGPU_SCAN_OUT_TRUNCATED
        q.put((rank, bool(ok), len(packed)))
    finally:
        dist.destroy_process_group()


def test_c5_one_sensor_group_per_rank_fused_message_world2_gloo():
    world, per_rank, n = 2, 2, 1200
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_c5_worker, args=(r, world, port, per_rank, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res
    assert res[0][2] == res[1][2] > 0
