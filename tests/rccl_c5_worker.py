"""Worker of tests/test_gpu2_rccl.py (one process per rank, started by torch.distributed.run):
BASELINE config 5 across ranks through the library's own RCCL exchange — every rank voxelises ITS
sensors' scans (E5 + E4), the clouds meet on every rank (META + 16-byte points, in-place all-gather
on the exchange stream), are unpacked, moved into the common frame and wrapped as ONE serialised
PointCloud2.  Every rank then repeats the chain WITHOUT the exchange on all sensors and compares:
the two messages must be the same bytes.  Prints RCCL_C5_OK <ranks> on rank 0."""
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))
from rplidar_ros2_driver_amd import Params, RplGpu, abi, synth  # noqa: E402

FID = "base_link"


def arena_of(gpu, dev, batch, p, cap):
    S, n = batch.shape
    d_nodes = torch.from_numpy(batch.view(np.uint8).reshape(S, n * 8)).to(dev)
    d_len = torch.full((S,), n, dtype=torch.int32, device=dev)
    t = dict(arena=torch.zeros(cap, 4, dtype=torch.float32, device=dev),
             cur=torch.zeros(1, dtype=torch.int64, device=dev),
             start=torch.zeros(S, dtype=torch.int64, device=dev),
             npts=torch.zeros(S, dtype=torch.int32, device=dev),
             st=torch.zeros(S, dtype=torch.int32, device=dev))
    gpu.cloud_arena_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), S, p, t["arena"].data_ptr(), cap,
                        t["cur"].data_ptr(), t["start"].data_ptr(), t["npts"].data_ptr(),
                        t["st"].data_ptr())
    return t


def main():
    import fusion_oracle as fo
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    S, n = 8, 16000
    assert S % world == 0
    per = S // world  # sensors of a rank
    batch = np.stack([synth.make_scan(800 + s, 0, n, noise_m=0.01) for s in range(S)])
    p = Params.defaults(clip_enable=1, range_max=40.0, ror_enable=1, voxel_enable=1)
    poses = np.stack([fo.planar_pose(0.7 * s - 1.0, 0.35 * s, -0.2 * s, 0.05 * s) for s in range(S)])
    with RplGpu(device=local, max_samples_per_scan=32768, max_batch=S) as gpu:
        stream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(stream)
        gpu.set_stream(stream.cuda_stream)
        uid = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            uid.copy_(torch.from_numpy(RplGpu.comm_unique_id()))
        dist.broadcast(uid, src=0)
        gpu.comm_init(rank, world, uid.cpu().numpy())
        assert gpu.comm_size() == (world, rank), gpu.comm_size()
        slot = per * 8192
        mine = arena_of(gpu, dev, batch[rank * per: (rank + 1) * per], p, slot)
        mw = abi.cloud_meta_words(per)
        d_meta = torch.zeros(mw, dtype=torch.int32, device=dev)
        gpu.pack_cloud_meta_dev(mine["cur"].data_ptr(), mine["start"].data_ptr(), mine["npts"].data_ptr(),
                                per, slot, per, d_meta.data_ptr())
        d_pts_all = torch.full((world, slot, 4), -9.0, dtype=torch.float32, device=dev)
        d_meta_all = torch.zeros(world, mw, dtype=torch.int32, device=dev)
        gpu.allgather_clouds_dev(mine["arena"].data_ptr(), slot, d_meta.data_ptr(), mw,
                                 d_pts_all.data_ptr(), d_meta_all.data_ptr())
        gpu.comm_fence()
        cap = world * slot
        d_packed = torch.zeros(cap, 4, dtype=torch.float32, device=dev)
        d_total = torch.zeros(1, dtype=torch.int64, device=dev)
        d_sa = torch.zeros(world, per, dtype=torch.int64, device=dev)
        d_na = torch.zeros(world, per, dtype=torch.int32, device=dev)
        d_stat = torch.zeros(1, dtype=torch.int32, device=dev)
        gpu.unpack_gathered_dev(d_pts_all.data_ptr(), slot, d_meta_all.data_ptr(), mw, world, per,
                                d_packed.data_ptr(), d_total.data_ptr(), d_sa.data_ptr(),
                                d_na.data_ptr(), d_stat.data_ptr())
        d_pose = torch.from_numpy(poses.reshape(S, 12)).to(dev)
        gpu.transform_clouds_dev(d_packed.data_ptr(), 0, d_sa.data_ptr(), d_na.data_ptr(), S,
                                 d_pose.data_ptr())
        msg_cap = abi.msg_cloud_layout(len(FID), cap).total_len
        d_msg = torch.zeros(msg_cap, dtype=torch.uint8, device=dev)
        d_ml = torch.zeros(1, dtype=torch.int64, device=dev)
        d_ms = torch.zeros(1, dtype=torch.int32, device=dev)
        gpu.fused_cloud_msg_dev(d_packed.data_ptr(), d_total.data_ptr(), cap, FID, 5, 6,
                                d_msg.data_ptr(), msg_cap, d_ml.data_ptr(), d_ms.data_ptr())
        gpu.synchronize()
        assert int(mine["st"].max()) == 0 and int(d_stat.item()) == 0 and int(d_ms.item()) == 0
        # every sensor on this rank alone, no exchange: the arena holds the scans in COMPLETION
        # order, the gathered cloud in (rank, completion) order — compare sensor by sensor
        full = arena_of(gpu, dev, batch, p, S * 8192)
        gpu.transform_clouds_dev(full["arena"].data_ptr(), 0, full["start"].data_ptr(),
                                 full["npts"].data_ptr(), S, d_pose.data_ptr())
        gpu.synchronize()
        fa, fs, fn_ = full["arena"].cpu().numpy(), full["start"].cpu().numpy(), full["npts"].cpu().numpy()
        ga, gs, gn = d_packed.cpu().numpy(), d_sa.cpu().numpy().reshape(-1), d_na.cpu().numpy().reshape(-1)
        assert int(d_total.item()) == int(fn_.sum()) == int(gn.sum())
        for s in range(S):
            assert int(gn[s]) == int(fn_[s]), s
            assert ga[gs[s]: gs[s] + gn[s]].tobytes() == fa[fs[s]: fs[s] + fn_[s]].tobytes(), s
        # the message: header + exactly the gathered cloud
        import cdr_oracle as cdr
        ml = int(d_ml.item())
        assert d_msg.cpu().numpy()[:ml].tobytes() == cdr.cloud_msg(FID, 5, 6, ga[: int(d_total.item())])
        # gather to ONE rank (rplgpu_gather_clouds_dev, grouped ncclSend / ncclRecv): the last rank is
        # the root; what it receives must be what the all-gather gave it, the others pass no buffers
        root = world - 1
        g_pts = torch.full((world, slot, 4), -5.0, dtype=torch.float32, device=dev) if rank == root else None
        g_meta = torch.zeros(world, mw, dtype=torch.int32, device=dev) if rank == root else None
        gpu.gather_clouds_dev(root, mine["arena"].data_ptr(), slot, 4, d_meta.data_ptr(), mw,
                              g_pts.data_ptr() if rank == root else 0,
                              g_meta.data_ptr() if rank == root else 0)
        gpu.comm_fence()
        gpu.synchronize()
        if rank == root:
            assert torch.equal(g_meta, d_meta_all)
            for r in range(world):
                k = int(d_meta_all[r, 0].item())
                assert torch.equal(g_pts[r, :k], d_pts_all[r, :k]), r
        gpu.comm_destroy()
    ok = torch.ones(1, dtype=torch.int32, device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if rank == 0 and int(ok.item()) == 1:
        print(f"RCCL_C5_OK {world}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
