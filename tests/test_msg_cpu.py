"""CPU tests of the serialised-message stage (include/rplgpu_msg.h):
the host-side CDR framing against hand-derived known-answer bytes and against the independent
restatement of the format in oracle/cdr_oracle.py.  No device is needed for any of this.

The wire format itself is "parity unpinned" (Fast-CDR / rosidl are not in the reference tree
nor in this image, see oracle/cdr_oracle.py); the known-answer bytes below were written out by
hand from the rules in include/rplgpu_msg.h.
"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import pytest

from rplidar_ros2_driver_amd import Params, ScanMeta, abi

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
import cdr_oracle as cdr  # noqa: E402

KAT_LASERSCAN = bytes.fromhex(
    "00010000"              # encapsulation: CDR little endian
    "01000000" "02000000"   # stamp.sec = 1, stamp.nanosec = 2
    "06000000" "6c6173657200" "0000"   # "laser" + NUL, 2 bytes of padding
    "00000000"              # angle_min 0.0
    "db0fc940"              # angle_max 6.2831855f
    "db0f4940"              # angle_increment 3.1415927f
    "cdcc4c3d"              # time_increment 0.05f
    "cdcccc3d"              # scan_time 0.1f
    "9a99193e"              # range_min 0.15f
    "00004041"              # range_max 12.0f
    "02000000" "0000803f" "00000040"   # ranges = [1, 2]
    "02000000" "00003c42" "00000000"   # intensities = [47, 0]
)

KAT_CLOUD = bytes.fromhex(
    "00010000"
    "01000000" "02000000"
    "02000000" "6c00" "0000"          # frame_id "l"
    "01000000"                         # height
    "01000000"                         # width
    "04000000"                         # fields.size()
    "02000000" "7800" "0000" "00000000" "07" "000000" "01000000"   # x, offset 0, FLOAT32, count 1
    "02000000" "7900" "0000" "04000000" "07" "000000" "01000000"   # y
    "02000000" "7a00" "0000" "08000000" "07" "000000" "01000000"   # z
    "0a000000" "696e74656e7369747900" "0000" "0c000000" "07" "000000" "01000000"  # intensity
    "00" "000000"                      # is_bigendian
    "10000000"                         # point_step
    "10000000"                         # row_step
    "10000000"                         # data.size()
    "0000803f" "00000040" "00000000" "00003c42"   # (1, 2, 0, 47)
    "01"                               # is_dense
)


def _meta(**kw):
    m = ScanMeta()
    for k, v in kw.items():
        setattr(m, k, v)
    return m


def test_known_answer_laserscan():
    m = _meta(angle_min=0.0, angle_max=np.float32(2 * np.pi), angle_increment=np.float32(np.pi),
              time_increment=0.05, scan_time=0.1, range_min=0.15, range_max=12.0, count=2,
              published=1)
    r = np.array([1.0, 2.0], np.float32)
    q = np.array([47.0, 0.0], np.float32)
    assert cdr.laserscan_msg("laser", 1, 2, m, r, q) == KAT_LASERSCAN
    buf = np.full(len(KAT_LASERSCAN) + 16, 0xEE, np.uint8)
    L = abi.msg_laserscan_header("laser", 1, 2, m, buf)
    assert (L.scalars_off, L.ranges_len_off, L.ranges_off, L.intensities_len_off,
            L.intensities_off, L.total_len) == (24, 52, 56, 64, 68, 76)
    buf[L.ranges_off: L.ranges_off + 8] = r.view(np.uint8)
    buf[L.intensities_off: L.intensities_off + 8] = q.view(np.uint8)
    assert bytes(buf[: L.total_len]) == KAT_LASERSCAN
    assert np.all(buf[L.total_len:] == 0xEE)  # nothing written past the message
    back = cdr.deserialize("LaserScan", KAT_LASERSCAN)
    assert back["header"]["frame_id"] == "laser" and back["ranges"].tolist() == [1.0, 2.0]


def test_known_answer_cloud():
    pts = np.array([[1.0, 2.0, 0.0, 47.0]], np.float32)
    assert cdr.cloud_msg("l", 1, 2, pts) == KAT_CLOUD
    buf = np.full(len(KAT_CLOUD) + 16, 0xEE, np.uint8)
    L = abi.msg_cloud_header("l", 1, 2, 1, buf)
    assert L.total_len == len(KAT_CLOUD) and L.is_dense_off == L.total_len - 1
    assert L.data_off == L.data_len_off + 4 and L.data_off % 4 == 0
    buf[L.data_off: L.data_off + 16] = pts.view(np.uint8).reshape(-1)
    assert bytes(buf[: L.total_len]) == KAT_CLOUD
    assert np.all(buf[L.total_len:] == 0xEE)
    back = cdr.deserialize("PointCloud2", KAT_CLOUD)
    assert [f["name"] for f in back["fields"]] == ["x", "y", "z", "intensity"]
    assert [f["offset"] for f in back["fields"]] == [0, 4, 8, 12]
    assert back["point_step"] == 16 and back["row_step"] == 16 and back["is_dense"] is True


@pytest.mark.parametrize("fid_len", list(range(0, 18)) + [63, 64, 255, 1000])
def test_headers_match_restatement_for_every_padding_case(fid_len):
    rng = np.random.default_rng(fid_len)
    fid = "".join(chr(97 + int(c)) for c in rng.integers(0, 26, fid_len))
    lib = abi.load_library()
    for count in (0, 1, 2, 3, 360, 4097):
        p = Params.defaults(range_max=40.0, scan_processing=int(count % 2))
        m = ScanMeta()
        lib.rplgpu_fill_meta(C.byref(p), count, 0.0731, C.byref(m))
        m.count = count  # count 0 is never published; the layout must still be right
        r = rng.random(count, dtype=np.float32)
        q = rng.random(count, dtype=np.float32)
        want = cdr.laserscan_msg(fid, -5, 999999999, m, r, q)
        buf = np.zeros(len(want), np.uint8)
        L = abi.msg_laserscan_header(fid, -5, 999999999, m, buf)
        assert L.total_len == len(want)
        assert all(getattr(L, f[0]) % 4 == 0 for f in L._fields_)
        assert bytes(abi.msg_laserscan_layout(fid_len, count)) == bytes(L)
        buf[L.ranges_off: L.ranges_off + 4 * count] = r.view(np.uint8)
        buf[L.intensities_off: L.intensities_off + 4 * count] = q.view(np.uint8)
        assert bytes(buf) == want
        back = cdr.deserialize("LaserScan", bytes(buf))
        assert back["header"] == {"stamp": {"sec": -5, "nanosec": 999999999}, "frame_id": fid}
        assert back["intensities"].tobytes() == q.tobytes()

        pts = rng.random((count, 4), dtype=np.float32)
        want = cdr.cloud_msg(fid, 2**31 - 1, 0, pts)
        buf = np.zeros(len(want), np.uint8)
        Lc = abi.msg_cloud_header(fid, 2**31 - 1, 0, count, buf)
        assert Lc.total_len == len(want) and Lc.data_off % 4 == 0
        assert bytes(abi.msg_cloud_layout(fid_len, count)) == bytes(Lc)
        buf[Lc.data_off: Lc.data_off + 16 * count] = pts.view(np.uint8).reshape(-1)
        assert bytes(buf) == want
        back = cdr.deserialize("PointCloud2", bytes(buf))
        assert back["width"] == count and back["data"].tobytes() == pts.tobytes()


def test_header_argument_errors():
    lib = abi.load_library()
    m = _meta(count=10, published=1)
    buf = np.zeros(64, np.uint8)
    L = abi.LaserScanLayout()
    st = abi.Stamp(0, 0)
    # too small: nothing is written, the layout is still reported
    rc = lib.rplgpu_msg_laserscan_header(b"laser", st, C.byref(m), buf.ctypes.data, 64, C.byref(L))
    assert rc == abi.ERR_CAPACITY and L.total_len == 24 + 28 + 4 + 40 + 4 + 40
    assert not buf.any()
    Lc = abi.CloudLayout()
    rc = lib.rplgpu_msg_cloud_header(b"laser", st, 10, buf.ctypes.data, 64, C.byref(Lc))
    assert rc == abi.ERR_CAPACITY and not buf.any()
    assert lib.rplgpu_msg_laserscan_header(None, st, C.byref(m), buf.ctypes.data, 64, None) \
        == abi.ERR_INVALID_ARG
    assert lib.rplgpu_msg_cloud_header(b"x", st, 1, None, 64, None) == abi.ERR_INVALID_ARG
    assert lib.rplgpu_msg_cloud_layout(4, 1 << 28, C.byref(Lc)) == abi.ERR_INVALID_ARG
    # handle-taking entry points reject a null handle before touching HIP
    assert lib.rplgpu_host_alloc(None, 16, C.byref(C.c_void_p())) == abi.ERR_INVALID_ARG
    ln = C.c_size_t(0)
    assert lib.rplgpu_scan_to_laserscan_msg(None, None, 0, None, 0.1, b"x", st, None, 0,
                                            C.byref(ln), None) == abi.ERR_INVALID_ARG
    assert lib.rplgpu_laserscan_msgs_dev(None, None, None, 0, None, 0, None, b"x", None, None,
                                         None, 0, None, None) == abi.ERR_INVALID_ARG
    assert lib.rplgpu_cloud_msgs_dev(None, None, 0, None, None, 0, b"x", None, None, 0, None,
                                     None) == abi.ERR_INVALID_ARG
