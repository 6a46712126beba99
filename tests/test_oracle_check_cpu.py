"""The whole-batch cloud checker of the oracle (orc_batch_cloud_check, used by tests/test_gpu_scale.py
to compare EVERY scan of a bench-sized batch): fed the oracle's own clouds it reports nothing, and
it sees a flipped cell key, a changed intensity, a moved point, a wrong z and a wrong count."""
import numpy as np

from rplidar_ros2_driver_amd import Params, synth
from tests import oracle_lib


def _own_output(oracle, batch, op):
    outs = [oracle.cloud_pipeline(batch[b], op) for b in range(len(batch))]
    npts = np.array([len(x[0]) for x in outs], np.uint32)
    start = (np.cumsum(npts) - npts).astype(np.uint64)
    arena = np.concatenate([x[0] for x in outs])
    keys = np.concatenate([((x[1][:, 1].astype(np.int64) + 32768) << 16) | (x[1][:, 0].astype(np.int64) + 32768)
                           for x in outs]).astype(np.uint32)
    return arena, start, npts, keys


def test_batch_cloud_check_accepts_the_oracle_and_sees_every_kind_of_difference(oracle):
    B, n = 12, 3000
    batch = synth.make_batch(7, B, n)
    p = Params.defaults(clip_enable=1, q_min=0, range_min=0.15, range_max=40.0, voxel_enable=1,
                        voxel_leaf=0.05)
    op = oracle_lib.copy_params(p)
    arena, start, npts, keys = _own_output(oracle, batch, op)
    bad, res = oracle.batch_cloud_check(batch, op, arena, start, npts, keys, 4)
    assert bad == 0 and np.array_equal(res[:, 0], npts) and not res[:, 1:].any()
    # scans in another order in the arena (completion order on the device): still fine
    order = np.arange(B)[::-1]
    start2 = np.zeros(B, np.uint64)
    pieces, kpieces, at = [], [], 0
    for b in order:
        start2[b] = at
        pieces.append(arena[start[b]: start[b] + npts[b]])
        kpieces.append(keys[start[b]: start[b] + npts[b]])
        at += int(npts[b])
    bad, _ = oracle.batch_cloud_check(batch, op, np.concatenate(pieces), start2, npts, np.concatenate(kpieces), 4)
    assert bad == 0
    for what in ("key", "intensity", "z", "xy", "count"):
        a, k, m = arena.copy(), keys.copy(), npts.copy()
        i = int(start[3]) + 5
        if what == "key":
            k[i] ^= 1
        elif what == "intensity":
            a[i, 3] = np.nextafter(a[i, 3], np.float32(1e9))
        elif what == "z":
            a[i, 2] = 1e-30
        elif what == "xy":
            a[i, 0] += np.float32(1e-3)
        else:
            m[3] -= 1
        bad, res = oracle.batch_cloud_check(batch, op, a, start, m, k, 4)
        if what == "xy":  # a moved point is reported through the distance, not as a mismatch
            assert bad == 0 and res[3, 3:4].copy().view(np.float32)[0] > 5e-4
        else:
            assert bad == 1 and (res[3, 1] or res[3, 2]), what
            assert not res[np.arange(B) != 3, 1:3].any()
