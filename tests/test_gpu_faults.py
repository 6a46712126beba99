"""Fault injection (a GPU hiccup must surface as a return code so that the node
falls back to its CPU loop instead of entering RESETTING, src/rplidar_node.cpp:453-474).
Every entry point is fed wrong arguments — null and HOST pointers where device memory is
expected, sizes beyond the handle's capacity, a too small arena, a wrong device id — and must
answer with a negative code (never crash, never hang); the very next valid call on the same
handle must still be correct."""
import ctypes as C

import numpy as np
import pytest

from rplidar_ros2_driver_amd import NODE_DTYPE, Params, RplGpu, abi, synth
from tests import oracle_lib

pytestmark = pytest.mark.gpu


def _still_works(gpu, oracle):
    nodes = synth.make_scan(7, 0, 2000)
    p = Params.defaults(range_max=40.0)
    r, i, m = gpu.scan_to_laserscan(nodes, p, 0.1)
    wr, wi, wm = oracle.publish_scan(nodes, oracle_lib.copy_params(p), 0.1)
    assert bytes(m) == bytes(wm) and r.tobytes() == wr.tobytes() and i.tobytes() == wi.tobytes()
    pv = Params.defaults(clip_enable=1, range_max=40.0, voxel_enable=1)
    cloud, status = gpu.scan_to_cloud(nodes, pv)
    want, _, _ = oracle.cloud_pipeline(nodes, oracle_lib.copy_params(pv))
    assert status == 0 and len(cloud) == len(want)
    assert cloud[:, 3].tobytes() == want[:, 3].tobytes()


def test_create_rejects_bad_arguments():
    lib = abi.load_library()
    h = C.c_void_p()
    assert lib.rplgpu_create(99, 8192, 4, C.byref(h)) == abi.ERR_NO_DEVICE
    assert lib.rplgpu_create(-1, 8192, 4, C.byref(h)) == abi.ERR_NO_DEVICE
    assert lib.rplgpu_create(0, 0, 4, C.byref(h)) == abi.ERR_INVALID_ARG
    assert lib.rplgpu_create(0, 40000, 4, C.byref(h)) == abi.ERR_INVALID_ARG
    assert lib.rplgpu_create(0, 8192, 0, C.byref(h)) == abi.ERR_INVALID_ARG
    assert lib.rplgpu_create(0, 8192, 4, None) == abi.ERR_INVALID_ARG
    assert not h.value


def test_batch_entry_points_reject_wrong_pointers_and_recover(oracle):
    import torch
    dev = torch.device("cuda:0")
    lib = abi.load_library()
    with RplGpu(device=0, max_samples_per_scan=8192, max_batch=8) as gpu:
        h = gpu._h
        B, n = 4, 2000
        batch = synth.make_batch(3, B, n)
        d_nodes = torch.from_numpy(batch.view(np.uint8).reshape(B, n * 8)).to(dev)
        d_len = torch.full((B,), n, dtype=torch.int32, device=dev)
        d_r = torch.zeros(B, n, dtype=torch.float32, device=dev)
        d_i = torch.zeros(B, n, dtype=torch.float32, device=dev)
        d_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
        d_st = torch.zeros(B, dtype=torch.int32, device=dev)
        p = Params.defaults(range_max=40.0)
        pv = Params.defaults(clip_enable=1, range_max=40.0, voxel_enable=1)
        host_nodes = np.ascontiguousarray(batch)  # plain host memory where device memory belongs
        host_len = np.full(B, n, np.int32)
        bad_args = [  # (d_nodes, d_len)
            (0, d_len.data_ptr()),
            (d_nodes.data_ptr(), 0),
            (host_nodes.ctypes.data, d_len.data_ptr()),
            (d_nodes.data_ptr(), host_len.ctypes.data),
        ]
        for dn, dl in bad_args:
            assert lib.rplgpu_laserscan_batch_dev(h, dn, n, dl, B, C.byref(p), d_r.data_ptr(),
                                                  d_i.data_ptr(), d_cnt.data_ptr()) == abi.ERR_INVALID_ARG
            assert lib.rplgpu_ascend_batch_dev(h, dn, n, dl, B, d_st.data_ptr()) == abi.ERR_INVALID_ARG
            assert lib.rplgpu_cloud_batch_dev(h, dn, n, dl, B, C.byref(pv), d_r.data_ptr(), n // 4,
                                              d_cnt.data_ptr(), d_st.data_ptr()) == abi.ERR_INVALID_ARG
            assert lib.rplgpu_last_error(h)
            _still_works(gpu, oracle)
        # capacity: more scans than the handle was created for; more samples than it stages
        assert lib.rplgpu_laserscan_batch_dev(h, d_nodes.data_ptr(), n, d_len.data_ptr(), 9,
                                              C.byref(p), d_r.data_ptr(), d_i.data_ptr(),
                                              d_cnt.data_ptr()) == abi.ERR_CAPACITY
        big = synth.make_scan(1, 0, 9000)
        out = np.zeros(9000, np.float32)
        meta = abi.ScanMeta()
        assert lib.rplgpu_scan_to_laserscan(h, big.ctypes.data, 9000, C.byref(p), 0.1,
                                            out.ctypes.data, out.ctypes.data,
                                            C.byref(meta)) == abi.ERR_CAPACITY
        res = C.c_uint32(0)
        assert lib.rplgpu_ascend(h, big.ctypes.data, 9000, C.byref(res)) == abi.ERR_CAPACITY
        _still_works(gpu, oracle)
        # null outputs / null parameters
        assert lib.rplgpu_laserscan_batch_dev(h, d_nodes.data_ptr(), n, d_len.data_ptr(), B, None,
                                              d_r.data_ptr(), d_i.data_ptr(),
                                              d_cnt.data_ptr()) == abi.ERR_INVALID_ARG
        assert lib.rplgpu_laserscan_batch_dev(h, d_nodes.data_ptr(), n, d_len.data_ptr(), B,
                                              C.byref(p), 0, d_i.data_ptr(),
                                              d_cnt.data_ptr()) == abi.ERR_INVALID_ARG
        assert lib.rplgpu_scan_to_cloud(h, None, 10, C.byref(pv), out.ctypes.data, C.byref(res),
                                        None) == abi.ERR_INVALID_ARG
        # parameters out of their domain
        pbad = Params.defaults(clip_enable=1, voxel_enable=1, voxel_leaf=0.0)
        assert lib.rplgpu_cloud_batch_dev(h, d_nodes.data_ptr(), n, d_len.data_ptr(), B,
                                          C.byref(pbad), d_r.data_ptr(), n // 4, d_cnt.data_ptr(),
                                          d_st.data_ptr()) == abi.ERR_INVALID_ARG
        pbad = Params.defaults(clip_enable=1, ror_enable=1, ror_radius=-1.0)
        assert lib.rplgpu_cloud_batch_dev(h, d_nodes.data_ptr(), n, d_len.data_ptr(), B,
                                          C.byref(pbad), d_r.data_ptr(), n // 4, d_cnt.data_ptr(),
                                          d_st.data_ptr()) == abi.ERR_INVALID_ARG
        _still_works(gpu, oracle)
        # a too small arena: flagged and clamped, nothing written past it, handle intact
        d_arena = torch.full((64 + 8, 4), -1.0, dtype=torch.float32, device=dev)
        d_cur = torch.zeros(1, dtype=torch.int64, device=dev)
        d_start = torch.zeros(B, dtype=torch.int64, device=dev)
        assert lib.rplgpu_cloud_arena_dev(h, d_nodes.data_ptr(), n, d_len.data_ptr(), B,
                                          C.byref(pv), d_arena.data_ptr(), 64, d_cur.data_ptr(),
                                          d_start.data_ptr(), d_cnt.data_ptr(),
                                          d_st.data_ptr()) == abi.OK
        gpu.synchronize()
        assert int(d_cur.item()) > 64 and bool((d_st.cpu().numpy() & abi.SCAN_OUT_TRUNCATED).any())
        assert int(d_cnt.sum().item()) <= 64
        assert np.all(d_arena.cpu().numpy()[64:] == -1.0)
        _still_works(gpu, oracle)
        # a null handle never dereferences
        assert lib.rplgpu_synchronize(None) == abi.ERR_INVALID_ARG
        assert lib.rplgpu_set_stream(None, None) == abi.ERR_INVALID_ARG
        lib.rplgpu_fill_meta(None, 5, 0.1, C.byref(meta))  # tolerated: no parameters, no crash
        lib.rplgpu_fill_meta(C.byref(p), 5, 0.1, None)
        assert meta.count == 0


def test_switching_streams_between_launches_is_safe(oracle):
    """rplgpu_set_stream drains the old stream: the voxel kernel's scan queue and the staging
    are per handle, not per stream."""
    import torch
    dev = torch.device("cuda:0")
    B, n = 64, 8000
    batch = synth.make_batch(9, B, n)
    pv = Params.defaults(clip_enable=1, range_max=40.0, voxel_enable=1)
    with RplGpu(device=0, max_samples_per_scan=8192, max_batch=B) as gpu:
        d_nodes = torch.from_numpy(batch.view(np.uint8).reshape(B, n * 8)).to(dev)
        d_len = torch.full((B,), n, dtype=torch.int32, device=dev)
        s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        torch.cuda.synchronize()
        outs = []
        for st in (s1, s2, s1, None):
            gpu.set_stream(st.cuda_stream if st is not None else None)
            d_x = torch.zeros(B, 4096, 4, dtype=torch.float32, device=dev)
            d_np = torch.zeros(B, dtype=torch.int32, device=dev)
            d_st = torch.zeros(B, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            gpu.cloud_batch_dev(d_nodes.data_ptr(), n, d_len.data_ptr(), B, pv, d_x.data_ptr(), 4096,
                                d_np.data_ptr(), d_st.data_ptr())
            outs.append((d_x, d_np, d_st))
        gpu.synchronize()
        torch.cuda.synchronize()
        ref = [t.cpu().numpy().tobytes() for t in outs[0]]
        for o in outs[1:]:
            assert [t.cpu().numpy().tobytes() for t in o] == ref
        want, _, _ = oracle.cloud_pipeline(batch[5], oracle_lib.copy_params(pv))
        assert int(outs[0][1][5]) == len(want)


def test_round2_entry_points_reject_bad_arguments_and_recover(oracle):
    """The entry points added in round 2 (decode -> scans, fused voxel grid, LaserScan -> cloud,
    the exchange) with wrong arguments: an error code, a message, and a handle that still works."""
    import torch
    from rplidar_ros2_driver_amd import capsules as cp
    dev = torch.device("cuda:0")
    lib = abi.load_library()
    with RplGpu(device=0, max_samples_per_scan=8192, max_batch=8) as gpu:
        h = gpu._h
        ans, nf, B = cp.ANS_DENSE_CAPSULED, 60, 3
        S, npf = cp.FRAME_SIZE[ans], cp.NODES_PER_FRAME[ans]
        streams = np.stack([cp.make_stream(ans, nf, 70 + b, payload="ring", frames_per_rev=9.0) for b in range(B)])
        d_bytes = torch.from_numpy(streams).to(dev)
        d_nf = torch.full((B,), nf, dtype=torch.int32, device=dev)
        scan_cap, n_stride = 8, 1024
        d_batch = torch.zeros(B * scan_cap, n_stride * 8, dtype=torch.uint8, device=dev)
        d_len = torch.zeros(B * scan_cap, dtype=torch.int32, device=dev)
        d_ns = torch.zeros(B, dtype=torch.int32, device=dev)
        d_st = torch.zeros(B, dtype=torch.int32, device=dev)

        def dec(ans_=ans, dur=125, bytes_=None, nfp=None, max_frames=nf, max_count=8192, batch=None,
                n_stride_=n_stride, cap=scan_cap, lens=None, st=None):
            return lib.rplgpu_decode_scans_dev(
                h, ans_, dur, d_bytes.data_ptr() if bytes_ is None else bytes_, nf * S, 0, 0,
                d_nf.data_ptr() if nfp is None else nfp, max_frames, B, 0, 0, max_count,
                d_batch.data_ptr() if batch is None else batch, n_stride_, cap,
                d_len.data_ptr() if lens is None else lens, d_ns.data_ptr(), 0,
                d_st.data_ptr() if st is None else st)

        assert dec() == abi.OK
        gpu.synchronize()
        good = (d_len.cpu().numpy().copy(), d_batch.cpu().numpy().copy())
        assert good[0].sum() > 0
        assert dec(ans_=0x42) == abi.ERR_INVALID_ARG
        assert dec(dur=0) == abi.ERR_INVALID_ARG
        assert dec(max_count=0) == abi.ERR_INVALID_ARG
        assert dec(cap=0) == abi.ERR_INVALID_ARG
        assert dec(n_stride_=0) == abi.ERR_INVALID_ARG
        assert dec(batch=0) == abi.ERR_INVALID_ARG
        assert dec(st=0) == abi.ERR_INVALID_ARG
        assert dec(max_frames=4096) == abi.ERR_CAPACITY
        assert dec(cap=70000) == abi.ERR_CAPACITY
        assert dec(bytes_=streams.ctypes.data) == abi.ERR_INVALID_ARG  # host memory
        assert lib.rplgpu_last_error(h)
        host_nf = np.full(B, nf, np.int32)
        assert dec(nfp=host_nf.ctypes.data) == abi.ERR_INVALID_ARG
        d_len.zero_(); d_batch.zero_()
        assert dec() == abi.OK
        gpu.synchronize()
        assert d_len.cpu().numpy().tobytes() == good[0].tobytes()
        assert d_batch.cpu().numpy().tobytes() == good[1].tobytes()
        _still_works(gpu, oracle)

        # fused voxel grid
        n, Bs = 1500, 4
        batch = synth.make_batch(5, Bs, n)
        d_nodes = torch.from_numpy(batch.view(np.uint8).reshape(Bs, n * 8)).to(dev)
        d_n = torch.full((Bs,), n, dtype=torch.int32, device=dev)
        d_arena = torch.zeros(Bs * n, 4, dtype=torch.float32, device=dev)
        d_cur = torch.zeros(1, dtype=torch.int64, device=dev)
        d_gs = torch.zeros(Bs, dtype=torch.int64, device=dev)
        d_np = torch.zeros(Bs, dtype=torch.int32, device=dev)
        pv = Params.defaults(clip_enable=1, range_max=40.0, voxel_enable=1)
        pn = Params.defaults(clip_enable=1, range_max=40.0, voxel_enable=0)

        def fused(group=2, p=pv, arena=None, motion=0):
            return lib.rplgpu_cloud_fused_voxel_dev(
                h, d_nodes.data_ptr(), n, d_n.data_ptr(), Bs, group, C.byref(p), motion, 0,
                d_arena.data_ptr() if arena is None else arena, Bs * n, d_cur.data_ptr(),
                d_gs.data_ptr(), d_np.data_ptr(), d_st.data_ptr())

        assert fused() == abi.OK
        assert fused(group=0) == abi.ERR_INVALID_ARG
        assert fused(p=pn) == abi.ERR_INVALID_ARG
        assert fused(arena=0) == abi.ERR_INVALID_ARG
        host_motion = np.zeros((Bs, 4), np.float32)
        assert fused(motion=host_motion.ctypes.data) == abi.ERR_INVALID_ARG
        # per-scan time offsets: a host pointer is refused; set, they require d_motion; cleared, all is as before
        host_t0 = np.zeros(Bs, np.float32)
        assert lib.rplgpu_set_scan_time_offsets_dev(h, host_t0.ctypes.data) == abi.ERR_INVALID_ARG
        assert lib.rplgpu_set_scan_time_offsets_dev(None, None) == abi.ERR_INVALID_ARG
        d_t0 = torch.zeros(Bs, dtype=torch.float32, device=dev)
        d_mo = torch.zeros(Bs, 4, dtype=torch.float32, device=dev)
        assert lib.rplgpu_set_scan_time_offsets_dev(h, d_t0.data_ptr()) == abi.OK
        assert fused() == abi.ERR_INVALID_ARG
        assert b"d_motion" in lib.rplgpu_last_error(h)
        assert fused(motion=d_mo.data_ptr()) == abi.OK
        assert lib.rplgpu_set_scan_time_offsets_dev(h, None) == abi.OK
        assert fused() == abi.OK
        _still_works(gpu, oracle)

        # LaserScan -> cloud, and the exchange before rplgpu_comm_init
        d_r = torch.zeros(Bs, n, dtype=torch.float32, device=dev)
        d_cnt = torch.full((Bs,), n, dtype=torch.int32, device=dev)
        assert lib.rplgpu_laserscan_to_cloud_batch_dev(h, 0, d_r.data_ptr(), n, d_cnt.data_ptr(), Bs,
                                                       C.byref(pn), d_arena.data_ptr(), n,
                                                       d_np.data_ptr(), d_st.data_ptr()) == abi.ERR_INVALID_ARG
        assert lib.rplgpu_laserscan_to_cloud_batch_dev(h, d_r.data_ptr(), d_r.data_ptr(), n,
                                                       d_cnt.data_ptr(), Bs, None, d_arena.data_ptr(), n,
                                                       d_np.data_ptr(), d_st.data_ptr()) == abi.ERR_INVALID_ARG
        d_meta = torch.zeros(64, dtype=torch.int32, device=dev)
        assert lib.rplgpu_allgather_clouds_dev(h, d_arena.data_ptr(), 16, d_meta.data_ptr(), 8,
                                               d_arena.data_ptr(), d_meta.data_ptr()) == abi.ERR_INVALID_ARG
        assert b"rplgpu_comm_init" in lib.rplgpu_last_error(h)
        assert lib.rplgpu_comm_fence(h) in (abi.OK, abi.ERR_INVALID_ARG)
        _still_works(gpu, oracle)
