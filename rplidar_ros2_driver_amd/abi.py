"""ctypes binding over the C ABI of ``librplgpu.so`` (``include/rplgpu.h``).

This is the same door the reference's C++ node would use; Python is
only plumbing for tests and ``bench.py``.  Nothing here computes on samples and nothing
falls back to a CPU implementation: a missing library raises ``RplGpuError``.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

# == sl_lidar_response_measurement_node_hq_t (reference src/sdk/include/sl_lidar_cmd.h:272-278)
NODE_DTYPE = np.dtype(
    [("angle_z_q14", "<u2"), ("dist_mm_q2", "<u4"), ("quality", "u1"), ("flag", "u1")]
)
assert NODE_DTYPE.itemsize == 8

MAX_SAMPLES_PER_SCAN = 32768

OK = 0
ERR_INVALID_ARG = -1
ERR_NO_DEVICE = -2
ERR_HIP = -3
ERR_CAPACITY = -4
ERR_ALL_INVALID = -5
ERR_SCAN_OVERFLOW = -6

SCAN_ALL_INVALID = 0x1
SCAN_CELL_RANGE = 0x2
SCAN_TABLE_FULL = 0x4
SCAN_OUT_TRUNCATED = 0x8

SL_RESULT_OK = 0
SL_RESULT_OPERATION_FAIL = 0x80008001

# every symbol include/rplgpu.h declares (checked by the CPU-side ABI test)
ABI_SYMBOLS = [
    "rplgpu_abi_version",
    "rplgpu_create",
    "rplgpu_destroy",
    "rplgpu_last_error",
    "rplgpu_default_params",
    "rplgpu_set_stream",
    "rplgpu_synchronize",
    "rplgpu_ascend",
    "rplgpu_scan_to_laserscan",
    "rplgpu_scan_to_cloud",
    "rplgpu_ascend_batch_dev",
    "rplgpu_laserscan_batch_dev",
    "rplgpu_ascend_laserscan_batch_dev",
    "rplgpu_cloud_batch_dev",
    "rplgpu_pack_clouds_dev",
    "rplgpu_cloud_arena_dev",
    "rplgpu_fill_meta",
    "rplgpu_frame_size",
    "rplgpu_nodes_per_frame",
    "rplgpu_decode_max_frames",
    "rplgpu_decode_staged_frames",
    "rplgpu_frame_stream",
    "rplgpu_decode_batch_dev",
    "rplgpu_decode_scans_dev",
    "rplgpu_decode_scans_carry_dev",
    "rplgpu_segment_batch_dev",
    "rplgpu_scans_to_batch_dev",
    "rplgpu_decode_stream",
    # include/rplgpu_msg.h
    "rplgpu_msg_laserscan_layout",
    "rplgpu_msg_cloud_layout",
    "rplgpu_msg_laserscan_header",
    "rplgpu_msg_cloud_header",
    "rplgpu_host_alloc",
    "rplgpu_host_free",
    "rplgpu_scan_to_laserscan_msg",
    "rplgpu_scan_to_cloud_msg",
    "rplgpu_laserscan_msgs_dev",
    "rplgpu_cloud_msgs_dev",
    "rplgpu_transform_clouds_dev",
    "rplgpu_fused_cloud_msg_dev",
    "rplgpu_cloud_deskew_batch_dev",
    "rplgpu_laserscan_to_cloud_batch_dev",
    "rplgpu_laserscan_to_cloud",
    "rplgpu_cloud_fused_voxel_dev",
    "rplgpu_set_cell_key_output",
    "rplgpu_set_scan_time_offsets_dev",
    "rplgpu_set_voxel_aggregation",
    "rplgpu_set_ror_mode",
    # include/rplgpu_comm.h
    "rplgpu_comm_unique_id",
    "rplgpu_comm_init",
    "rplgpu_comm_destroy",
    "rplgpu_comm_size",
    "rplgpu_cloud_meta_words",
    "rplgpu_pack_cloud_meta_dev",
    "rplgpu_allgather_clouds_dev",
    "rplgpu_gather_clouds_dev",
    "rplgpu_comm_fence",
    "rplgpu_comm_fence_lag",
    "rplgpu_unpack_gathered_dev",
    "rplgpu_pack_cloud_xyi_dev",
    "rplgpu_cloud_arena_xyi_dev",
    "rplgpu_allgather_clouds_xyi_dev",
    "rplgpu_unpack_gathered_xyi_dev",
    "rplgpu_pack_cloud_meta_host",
    "rplgpu_pack_cloud_xyi_host",
    "rplgpu_unpack_gathered_host",
]


class RplGpuError(RuntimeError):
    def __init__(self, code: int, msg: str = ""):
        super().__init__(f"rplgpu error {code}: {msg}")
        self.code = code


class Params(C.Structure):
    """Mirror of ``rplgpu_params_t``."""

    _fields_ = [
        ("is_new_protocol", C.c_int32),
        ("inverted", C.c_int32),
        ("scan_processing", C.c_int32),
        ("clip_enable", C.c_int32),
        ("q_min", C.c_uint32),
        ("range_min", C.c_float),
        ("range_max", C.c_float),
        ("voxel_leaf", C.c_float),
        ("ror_radius", C.c_float),
        ("ror_min_neighbors", C.c_uint32),
        ("ror_enable", C.c_int32),
        ("voxel_enable", C.c_int32),
    ]

    @classmethod
    def defaults(cls, **kw) -> "Params":
        p = cls(0, 0, 1, 0, 0, 0.15, 12.0, 0.05, 0.10, 2, 0, 0)
        for k, v in kw.items():
            if not hasattr(p, k):
                raise AttributeError(k)
            setattr(p, k, v)
        return p


class ScanMeta(C.Structure):
    """Mirror of ``rplgpu_scan_meta_t``."""

    _fields_ = [
        ("angle_min", C.c_float),
        ("angle_max", C.c_float),
        ("angle_increment", C.c_float),
        ("time_increment", C.c_float),
        ("scan_time", C.c_float),
        ("range_min", C.c_float),
        ("range_max", C.c_float),
        ("count", C.c_uint32),
        ("published", C.c_int32),
    ]


class Stamp(C.Structure):
    """Mirror of ``rplgpu_stamp_t`` (builtin_interfaces/Time)."""

    _fields_ = [("sec", C.c_int32), ("nanosec", C.c_uint32)]


class LaserScanLayout(C.Structure):
    """Mirror of ``rplgpu_laserscan_layout_t``."""

    _fields_ = [(k, C.c_uint32) for k in (
        "scalars_off", "ranges_len_off", "ranges_off", "intensities_len_off",
        "intensities_off", "total_len")]


class CloudLayout(C.Structure):
    """Mirror of ``rplgpu_cloud_layout_t``."""

    _fields_ = [(k, C.c_uint32) for k in (
        "width_off", "row_step_off", "data_len_off", "data_off", "is_dense_off", "total_len")]


def library_path() -> Path:
    env = os.environ.get("RPLGPU_LIBRARY")
    if env:
        return Path(env)
    return Path(__file__).resolve().parent / "lib" / "librplgpu.so"


_LIB = None


def load_library() -> C.CDLL:
    """Load ``librplgpu.so`` (in-tree build).  Raises if it is missing — no fallback."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not path.exists():
        raise RplGpuError(
            ERR_NO_DEVICE,
            f"{path} not built; run `python -c 'import __graft_entry__ as g; g.build()'`",
        )
    lib = C.CDLL(str(path))
    vp, u32, i32, sz = C.c_void_p, C.c_uint32, C.c_int32, C.c_size_t
    lib.rplgpu_abi_version.restype = i32
    lib.rplgpu_create.argtypes = [i32, u32, u32, C.POINTER(vp)]
    lib.rplgpu_destroy.argtypes = [vp]
    lib.rplgpu_last_error.argtypes = [vp]
    lib.rplgpu_last_error.restype = C.c_char_p
    lib.rplgpu_default_params.argtypes = [C.POINTER(Params)]
    lib.rplgpu_default_params.restype = None
    lib.rplgpu_set_stream.argtypes = [vp, vp]
    lib.rplgpu_set_voxel_aggregation.argtypes = [vp, i32]
    lib.rplgpu_set_ror_mode.argtypes = [vp, i32]
    lib.rplgpu_synchronize.argtypes = [vp]
    lib.rplgpu_ascend.argtypes = [vp, vp, sz, C.POINTER(u32)]
    lib.rplgpu_scan_to_laserscan.argtypes = [
        vp, vp, sz, C.POINTER(Params), C.c_double, vp, vp, C.POINTER(ScanMeta)]
    lib.rplgpu_scan_to_cloud.argtypes = [
        vp, vp, sz, C.POINTER(Params), vp, C.POINTER(u32), C.POINTER(u32)]
    lib.rplgpu_ascend_batch_dev.argtypes = [vp, vp, u32, vp, u32, vp]
    lib.rplgpu_laserscan_batch_dev.argtypes = [vp, vp, u32, vp, u32, C.POINTER(Params), vp, vp, vp]
    lib.rplgpu_ascend_laserscan_batch_dev.argtypes = [vp, vp, u32, vp, u32, C.POINTER(Params), vp, vp,
                                                      vp, C.c_int32, vp]
    lib.rplgpu_cloud_batch_dev.argtypes = [
        vp, vp, u32, vp, u32, C.POINTER(Params), vp, u32, vp, vp]
    lib.rplgpu_pack_clouds_dev.argtypes = [vp, vp, u32, vp, u32, vp, vp]
    lib.rplgpu_cloud_arena_dev.argtypes = [
        vp, vp, u32, vp, u32, C.POINTER(Params), vp, C.c_uint64, vp, vp, vp, vp]
    lib.rplgpu_cloud_arena_xyi_dev.argtypes = [
        vp, vp, u32, vp, u32, C.POINTER(Params), vp, C.c_uint64, vp, vp, vp, vp]
    lib.rplgpu_fill_meta.argtypes = [C.POINTER(Params), u32, C.c_double, C.POINTER(ScanMeta)]
    lib.rplgpu_fill_meta.restype = None
    u8, u64 = C.c_uint8, C.c_uint64
    lib.rplgpu_frame_size.argtypes = [u8]
    lib.rplgpu_frame_size.restype = sz
    lib.rplgpu_nodes_per_frame.argtypes = [u8]
    lib.rplgpu_nodes_per_frame.restype = sz
    lib.rplgpu_decode_max_frames.argtypes = [u8]
    lib.rplgpu_decode_max_frames.restype = u32
    lib.rplgpu_decode_staged_frames.argtypes = [u8]
    lib.rplgpu_decode_staged_frames.restype = u32
    lib.rplgpu_frame_stream.argtypes = [u8, vp, sz, vp, vp, sz]
    lib.rplgpu_frame_stream.restype = sz
    lib.rplgpu_decode_batch_dev.argtypes = [vp, u8, u32, vp, u64, vp, vp, vp, u32, u32, vp, vp,
                                            vp, u32, vp, vp, u32, vp, vp, vp]
    lib.rplgpu_decode_scans_dev.argtypes = [vp, u8, u32, vp, u64, vp, vp, vp, u32, u32, vp, vp,
                                            u32, vp, u32, u32, vp, vp, vp, vp]
    lib.rplgpu_decode_scans_carry_dev.argtypes = [vp, u8, u32, vp, u64, vp, vp, vp, u32, u32, vp, vp,
                                                  u32, vp, u32, u32, vp, vp, vp, vp, vp, vp, vp, vp, u32]
    lib.rplgpu_segment_batch_dev.argtypes = [vp, vp, u32, vp, vp, u32, vp, u32, u32, vp, u32, vp,
                                             u32, vp, vp]
    lib.rplgpu_scans_to_batch_dev.argtypes = [vp, vp, u32, vp, u32, vp, u32, vp, vp, u32, u32, vp]
    lib.rplgpu_decode_stream.argtypes = [vp, u8, u32, vp, sz, vp, vp, sz, C.POINTER(sz), vp, sz,
                                         C.POINTER(sz), C.POINTER(u32)]
    cs = C.c_char_p
    lib.rplgpu_msg_laserscan_layout.argtypes = [sz, u32, C.POINTER(LaserScanLayout)]
    lib.rplgpu_msg_cloud_layout.argtypes = [sz, u32, C.POINTER(CloudLayout)]
    lib.rplgpu_msg_laserscan_header.argtypes = [cs, Stamp, C.POINTER(ScanMeta), vp, sz,
                                                C.POINTER(LaserScanLayout)]
    lib.rplgpu_msg_cloud_header.argtypes = [cs, Stamp, u32, vp, sz, C.POINTER(CloudLayout)]
    lib.rplgpu_host_alloc.argtypes = [vp, sz, C.POINTER(vp)]
    lib.rplgpu_host_free.argtypes = [vp, vp]
    lib.rplgpu_scan_to_laserscan_msg.argtypes = [vp, vp, sz, C.POINTER(Params), C.c_double, cs,
                                                 Stamp, vp, sz, C.POINTER(sz), C.POINTER(ScanMeta)]
    lib.rplgpu_scan_to_cloud_msg.argtypes = [vp, vp, sz, C.POINTER(Params), cs, Stamp, vp, sz,
                                             C.POINTER(sz), C.POINTER(u32), C.POINTER(u32)]
    lib.rplgpu_laserscan_msgs_dev.argtypes = [vp, vp, vp, u32, vp, u32, C.POINTER(Params), cs,
                                              vp, vp, vp, u32, vp, vp]
    lib.rplgpu_cloud_msgs_dev.argtypes = [vp, vp, u32, vp, vp, u32, cs, vp, vp, u32, vp, vp]
    lib.rplgpu_transform_clouds_dev.argtypes = [vp, vp, u32, vp, vp, u32, vp]
    lib.rplgpu_fused_cloud_msg_dev.argtypes = [vp, vp, vp, u64, cs, Stamp, vp, u64, vp, vp]
    lib.rplgpu_cloud_deskew_batch_dev.argtypes = [
        vp, vp, u32, vp, u32, C.POINTER(Params), vp, vp, u32, vp, vp]
    lib.rplgpu_laserscan_to_cloud_batch_dev.argtypes = [vp, vp, vp, u32, vp, u32, C.POINTER(Params),
                                                        vp, u32, vp, vp]
    lib.rplgpu_laserscan_to_cloud.argtypes = [vp, vp, vp, u32, C.POINTER(Params), vp,
                                              C.POINTER(u32)]
    lib.rplgpu_cloud_fused_voxel_dev.argtypes = [vp, vp, u32, vp, u32, u32, C.POINTER(Params), vp, vp, vp,
                                                 u64, vp, vp, vp, vp]
    lib.rplgpu_set_cell_key_output.argtypes = [vp, vp]
    lib.rplgpu_set_scan_time_offsets_dev.argtypes = [vp, vp]
    lib.rplgpu_comm_unique_id.argtypes = [vp]
    lib.rplgpu_comm_init.argtypes = [vp, i32, i32, vp]
    lib.rplgpu_comm_destroy.argtypes = [vp]
    lib.rplgpu_comm_size.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    lib.rplgpu_cloud_meta_words.argtypes = [u32]
    lib.rplgpu_cloud_meta_words.restype = u32
    lib.rplgpu_pack_cloud_meta_dev.argtypes = [vp, vp, vp, vp, u32, u64, u32, vp]
    lib.rplgpu_allgather_clouds_dev.argtypes = [vp, vp, u64, vp, u32, vp, vp]
    lib.rplgpu_gather_clouds_dev.argtypes = [vp, i32, vp, u64, u32, vp, u32, vp, vp]
    lib.rplgpu_comm_fence.argtypes = [vp]
    lib.rplgpu_comm_fence_lag.argtypes = [vp, u32]
    lib.rplgpu_unpack_gathered_dev.argtypes = [vp, vp, u64, vp, u32, u32, u32, vp, vp, vp, vp, vp]
    lib.rplgpu_pack_cloud_xyi_dev.argtypes = [vp, vp, vp, u64, vp]
    lib.rplgpu_allgather_clouds_xyi_dev.argtypes = [vp, vp, u64, vp, u32, vp, vp]
    lib.rplgpu_unpack_gathered_xyi_dev.argtypes = [vp, vp, u64, vp, u32, u32, u32, vp, vp, vp, vp, vp]
    lib.rplgpu_pack_cloud_meta_host.argtypes = [u64, vp, vp, u32, u64, u32, vp]
    lib.rplgpu_pack_cloud_xyi_host.argtypes = [vp, u64, u64, vp]
    lib.rplgpu_unpack_gathered_host.argtypes = [vp, u64, u32, vp, u32, u32, u32, vp, vp, vp, vp, vp]
    for name in ABI_SYMBOLS:
        fn = getattr(lib, name)
        if fn.restype is C.c_int:  # default
            fn.restype = i32
    _LIB = lib
    return lib


def _nodes_ptr(arr: np.ndarray) -> int:
    if arr.dtype != NODE_DTYPE or not arr.flags["C_CONTIGUOUS"]:
        raise TypeError("nodes must be a C-contiguous array of NODE_DTYPE")
    return arr.ctypes.data


class RplGpu:
    """One ``rplgpu_handle_t`` (== one lidar node instance / one GPU stream)."""

    def __init__(self, device: int = 0, max_samples_per_scan: int = MAX_SAMPLES_PER_SCAN,
                 max_batch: int = 4096):
        self._lib = load_library()
        self._pinned = {}
        h = C.c_void_p()
        rc = self._lib.rplgpu_create(device, max_samples_per_scan, max_batch, C.byref(h))
        if rc != OK:
            msg = self._lib.rplgpu_last_error(None) or b""
            raise RplGpuError(rc, "rplgpu_create failed: " + msg.decode())
        self._h = h
        self.device = device
        self.max_samples_per_scan = max_samples_per_scan
        self.max_batch = max_batch

    # -- lifecycle -------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.rplgpu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc: int, allow=()):
        if rc != OK and rc not in allow:
            msg = self._lib.rplgpu_last_error(self._h) or b""
            raise RplGpuError(rc, msg.decode())
        return rc

    def set_stream(self, hip_stream_ptr: int | None):
        self._check(self._lib.rplgpu_set_stream(self._h, C.c_void_p(hip_stream_ptr or 0)))

    def synchronize(self):
        self._check(self._lib.rplgpu_synchronize(self._h))

    # -- single scan, host buffers -----------------------------------------------------
    def ascend(self, nodes: np.ndarray) -> int:
        """In place; returns the SDK ``sl_result`` (0 or 0x80008001)."""
        res = C.c_uint32(0)
        self._check(self._lib.rplgpu_ascend(self._h, _nodes_ptr(nodes), len(nodes), C.byref(res)))
        return res.value

    def scan_to_laserscan(self, nodes: np.ndarray, params: Params, scan_duration: float = 0.1):
        n = len(nodes)
        ranges = np.empty(max(n, 1), np.float32)
        intens = np.empty(max(n, 1), np.float32)
        meta = ScanMeta()
        self._check(self._lib.rplgpu_scan_to_laserscan(
            self._h, _nodes_ptr(nodes), n, C.byref(params), scan_duration,
            ranges.ctypes.data, intens.ctypes.data, C.byref(meta)))
        return ranges[: meta.count], intens[: meta.count], meta

    def scan_to_cloud(self, nodes: np.ndarray, params: Params, allow_overflow: bool = False):
        n = len(nodes)
        xyzi = np.empty((max(n, 1), 4), np.float32)
        npts = C.c_uint32(0)
        status = C.c_uint32(0)
        self._check(self._lib.rplgpu_scan_to_cloud(
            self._h, _nodes_ptr(nodes), n, C.byref(params), xyzi.ctypes.data,
            C.byref(npts), C.byref(status)),
            allow=(ERR_SCAN_OVERFLOW,) if allow_overflow else ())
        return xyzi[: npts.value], status.value

    # -- device-resident batches (raw device pointers, e.g. torch ``data_ptr()``) ---------
    def ascend_batch_dev(self, d_nodes: int, n_stride: int, d_n_per_scan: int, B: int,
                         d_status: int = 0):
        self._check(self._lib.rplgpu_ascend_batch_dev(
            self._h, d_nodes, n_stride, d_n_per_scan, B, d_status))

    def laserscan_batch_dev(self, d_nodes: int, n_stride: int, d_n_per_scan: int, B: int,
                            params: Params, d_ranges: int, d_intens: int, d_beam_count: int):
        self._check(self._lib.rplgpu_laserscan_batch_dev(
            self._h, d_nodes, n_stride, d_n_per_scan, B, C.byref(params),
            d_ranges, d_intens, d_beam_count))

    def ascend_laserscan_batch_dev(self, d_nodes: int, n_stride: int, d_n_per_scan: int, B: int,
                                   params: Params, d_ranges: int, d_intens: int, d_beam_count: int,
                                   write_ascended: bool = False, d_status: int = 0):
        """S1 -> S3 (grab_scan_data with geometric correction, then publish_scan) in one pass."""
        self._check(self._lib.rplgpu_ascend_laserscan_batch_dev(
            self._h, d_nodes, n_stride, d_n_per_scan, B, C.byref(params),
            d_ranges, d_intens, d_beam_count, int(bool(write_ascended)), d_status))

    def cloud_batch_dev(self, d_nodes: int, n_stride: int, d_n_per_scan: int, B: int,
                        params: Params, d_xyzi: int, out_stride: int, d_n_points: int,
                        d_status: int = 0):
        self._check(self._lib.rplgpu_cloud_batch_dev(
            self._h, d_nodes, n_stride, d_n_per_scan, B, C.byref(params),
            d_xyzi, out_stride, d_n_points, d_status))

    def cloud_arena_dev(self, d_nodes: int, n_stride: int, d_n_per_scan: int, B: int,
                        params: Params, d_arena: int, arena_capacity: int, d_cursor: int,
                        d_scan_start: int, d_n_points: int, d_status: int = 0):
        """Voxelised clouds of the batch in one contiguous arena (no packing pass)."""
        self._check(self._lib.rplgpu_cloud_arena_dev(
            self._h, d_nodes, n_stride, d_n_per_scan, B, C.byref(params), d_arena,
            arena_capacity, d_cursor, d_scan_start, d_n_points, d_status))

    def pack_clouds_dev(self, d_xyzi: int, out_stride: int, d_n_points: int, B: int,
                        d_packed: int, d_offsets: int):
        self._check(self._lib.rplgpu_pack_clouds_dev(
            self._h, d_xyzi, out_stride, d_n_points, B, d_packed, d_offsets))

    def fill_meta(self, params: Params, count: int, scan_duration: float) -> ScanMeta:
        meta = ScanMeta()
        self._lib.rplgpu_fill_meta(C.byref(params), count, scan_duration, C.byref(meta))
        return meta

    # -- serialised messages (include/rplgpu_msg.h) ----------------------
    def host_alloc(self, nbytes: int) -> np.ndarray:
        """Pinned host bytes (``rplgpu_host_alloc``) as a uint8 array; free with ``host_free``."""
        ptr = C.c_void_p()
        self._check(self._lib.rplgpu_host_alloc(self._h, nbytes, C.byref(ptr)))
        buf = (C.c_uint8 * nbytes).from_address(ptr.value)
        arr = np.frombuffer(buf, dtype=np.uint8)
        self._pinned[arr.ctypes.data] = ptr.value
        return arr

    def host_free(self, arr: np.ndarray):
        ptr = self._pinned.pop(arr.ctypes.data)
        self._check(self._lib.rplgpu_host_free(self._h, ptr))

    def scan_to_laserscan_msg(self, nodes: np.ndarray, params: Params, scan_duration: float,
                              frame_id: str, sec: int, nanosec: int, out: np.ndarray | None = None):
        """Serialised sensor_msgs/LaserScan of one scan: ``(bytes view, meta)``; the view is
        empty when publish_scan would not have published."""
        n = len(nodes)
        if out is None:
            out = np.empty(msg_laserscan_layout(len(frame_id.encode()), max(n, 1)).total_len,
                           np.uint8)
        meta = ScanMeta()
        ln = C.c_size_t(0)
        self._check(self._lib.rplgpu_scan_to_laserscan_msg(
            self._h, _nodes_ptr(nodes), n, C.byref(params), scan_duration, frame_id.encode(),
            Stamp(sec, nanosec), out.ctypes.data, out.nbytes, C.byref(ln), C.byref(meta)))
        return out[: ln.value], meta

    def scan_to_cloud_msg(self, nodes: np.ndarray, params: Params, frame_id: str, sec: int,
                          nanosec: int, out: np.ndarray | None = None,
                          allow_overflow: bool = False):
        """Serialised sensor_msgs/PointCloud2 of one scan: ``(bytes view, n_points, status)``."""
        n = len(nodes)
        if out is None:
            out = np.empty(msg_cloud_layout(len(frame_id.encode()), max(n, 1)).total_len, np.uint8)
        ln = C.c_size_t(0)
        npts, status = C.c_uint32(0), C.c_uint32(0)
        self._check(self._lib.rplgpu_scan_to_cloud_msg(
            self._h, _nodes_ptr(nodes), n, C.byref(params), frame_id.encode(),
            Stamp(sec, nanosec), out.ctypes.data, out.nbytes, C.byref(ln), C.byref(npts),
            C.byref(status)), allow=(ERR_SCAN_OVERFLOW,) if allow_overflow else ())
        return out[: ln.value], npts.value, status.value

    def laserscan_msgs_dev(self, d_ranges: int, d_intens: int, n_stride: int, d_beam_count: int,
                           B: int, params: Params, frame_id: str, d_stamps: int,
                           d_scan_duration: int, d_msgs: int, msg_stride: int, d_msg_len: int,
                           d_status: int = 0):
        self._check(self._lib.rplgpu_laserscan_msgs_dev(
            self._h, d_ranges, d_intens, n_stride, d_beam_count, B, C.byref(params),
            frame_id.encode(), d_stamps, d_scan_duration, d_msgs, msg_stride, d_msg_len, d_status))

    def cloud_msgs_dev(self, d_xyzi: int, out_stride: int, d_scan_start: int, d_n_points: int,
                       B: int, frame_id: str, d_stamps: int, d_msgs: int, msg_stride: int,
                       d_msg_len: int, d_status: int = 0):
        self._check(self._lib.rplgpu_cloud_msgs_dev(
            self._h, d_xyzi, out_stride, d_scan_start, d_n_points, B, frame_id.encode(),
            d_stamps, d_msgs, msg_stride, d_msg_len, d_status))

    def transform_clouds_dev(self, d_xyzi: int, out_stride: int, d_scan_start: int,
                             d_n_points: int, B: int, d_pose: int):
        """In-place rigid transform of B clouds, one row-major 3x4 float pose per scan."""
        self._check(self._lib.rplgpu_transform_clouds_dev(
            self._h, d_xyzi, out_stride, d_scan_start, d_n_points, B, d_pose))

    def cloud_deskew_batch_dev(self, d_nodes: int, n_stride: int, d_n_per_scan: int, B: int,
                               params: Params, d_motion: int, d_xyzi: int, out_stride: int,
                               d_n_points: int, d_status: int = 0):
        """Plain clouds with motion de-skew; d_motion: B x (vx, vy, wz, time_increment) float32."""
        self._check(self._lib.rplgpu_cloud_deskew_batch_dev(
            self._h, d_nodes, n_stride, d_n_per_scan, B, C.byref(params), d_motion, d_xyzi,
            out_stride, d_n_points, d_status))

    def laserscan_to_cloud_batch_dev(self, d_ranges: int, d_intens: int, n_stride: int,
                                     d_beam_count: int, B: int, params: Params, d_xyzi: int,
                                     out_stride: int, d_n_points: int, d_status: int = 0):
        """E7: the LaserScans of a batch projected to clouds (laser_geometry-style)."""
        self._check(self._lib.rplgpu_laserscan_to_cloud_batch_dev(
            self._h, d_ranges, d_intens, n_stride, d_beam_count, B, C.byref(params), d_xyzi,
            out_stride, d_n_points, d_status))

    def laserscan_to_cloud(self, ranges: np.ndarray, intens: np.ndarray, params: Params):
        """E7, one scan, host buffers: (count,) float32 ranges / intensities -> (m, 4) cloud."""
        ranges = np.ascontiguousarray(ranges, np.float32)
        intens = np.ascontiguousarray(intens, np.float32)
        count = len(ranges)
        xyzi = np.empty((max(count, 1), 4), np.float32)
        npts = C.c_uint32(0)
        self._check(self._lib.rplgpu_laserscan_to_cloud(
            self._h, ranges.ctypes.data, intens.ctypes.data, count, C.byref(params),
            xyzi.ctypes.data, C.byref(npts)))
        return xyzi[: npts.value]

    def cloud_fused_voxel_dev(self, d_nodes: int, n_stride: int, d_n_per_scan: int, B: int, group: int,
                              params: Params, d_motion: int, d_pose2d: int, d_arena: int,
                              arena_capacity: int, d_cursor: int, d_group_start: int, d_n_points: int,
                              d_status: int = 0):
        """E8: one voxel grid per group of `group` consecutive scans (de-skew + planar pose)."""
        self._check(self._lib.rplgpu_cloud_fused_voxel_dev(
            self._h, d_nodes, n_stride, d_n_per_scan, B, group, C.byref(params), d_motion, d_pose2d,
            d_arena, arena_capacity, d_cursor, d_group_start, d_n_points, d_status))

    def set_voxel_aggregation(self, mode: int = 0):
        """0 = auto (from the previous batch launch's statistics), 1 = plain, 2 = two-class
        (include/rplgpu.h RPLGPU_VOXEL_AGG_*).  Results are identical in every mode."""
        self._check(self._lib.rplgpu_set_voxel_aggregation(self._h, int(mode)))

    def set_ror_mode(self, mode: int = 0):
        """0 = E5 inside the voxel kernel (arena entry points, one pass over the scans), 1 = two kernels
        (include/rplgpu.h RPLGPU_ROR_*).  Results are identical in both modes."""
        self._check(self._lib.rplgpu_set_ror_mode(self._h, int(mode)))

    def debug_ror_listed(self) -> int:
        """Work items the last E5-inside launch left to the two kernels (waits for the stream)."""
        n = C.c_uint32(0)
        self._lib.rplgpu_debug_ror_listed.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        self._check(self._lib.rplgpu_debug_ror_listed(self._h, C.byref(n)))
        return int(n.value)

    def set_scan_time_offsets_dev(self, d_t0: int = 0):
        """Per scan of the following de-skew / fused launches: the time [s] of its first sample relative
        to the fused instant (include/rplgpu_msg.h); 0 switches the offsets off."""
        self._check(self._lib.rplgpu_set_scan_time_offsets_dev(self._h, d_t0 or None))

    def set_cell_key_output(self, d_cell_keys: int = 0):
        """Optional voxel output: one u32 per output point, (iy + 32768) << 16 | (ix + 32768)."""
        self._check(self._lib.rplgpu_set_cell_key_output(self._h, d_cell_keys or None))

    # -- multi-GPU exchange (include/rplgpu_comm.h) ------------------------------------------
    @staticmethod
    def comm_unique_id() -> np.ndarray:
        """A fresh RCCL unique id (rank 0 makes it, the caller carries it to the other ranks)."""
        uid = np.zeros(128, np.uint8)
        rc = load_library().rplgpu_comm_unique_id(uid.ctypes.data)
        if rc:
            raise RplGpuError(rc, "rplgpu_comm_unique_id (is librccl.so loadable?)")
        return uid

    def comm_init(self, rank: int, world: int, uid: np.ndarray):
        uid = np.ascontiguousarray(uid, np.uint8)
        assert uid.nbytes == 128
        self._check(self._lib.rplgpu_comm_init(self._h, rank, world, uid.ctypes.data))

    def comm_size(self):
        """(ranks, this rank) of the handle's communicator as RCCL reports them; (0, -1) without one."""
        w, r = C.c_int32(0), C.c_int32(-1)
        self._check(self._lib.rplgpu_comm_size(self._h, C.byref(w), C.byref(r)))
        return int(w.value), int(r.value)

    def comm_destroy(self):
        self._check(self._lib.rplgpu_comm_destroy(self._h))

    def pack_cloud_meta_dev(self, d_cursor: int, d_scan_start: int, d_n_points: int, B: int,
                            slot_points: int, max_scans: int, d_meta: int):
        self._check(self._lib.rplgpu_pack_cloud_meta_dev(
            self._h, d_cursor, d_scan_start, d_n_points, B, slot_points, max_scans, d_meta))

    def allgather_clouds_dev(self, d_points_local: int, slot_points: int, d_meta_local: int,
                             meta_words: int, d_points_all: int, d_meta_all: int):
        self._check(self._lib.rplgpu_allgather_clouds_dev(
            self._h, d_points_local, slot_points, d_meta_local, meta_words, d_points_all, d_meta_all))

    def cloud_arena_xyi_dev(self, d_nodes: int, n_stride: int, d_n_per_scan: int, B: int, p: Params,
                            d_slot: int, slot_points: int, d_cursor: int, d_scan_start: int,
                            d_n_points: int, d_status: int = 0):
        """cloud_arena_dev writing 12-byte points (x, y, intensity) straight into an exchange slot."""
        self._check(self._lib.rplgpu_cloud_arena_xyi_dev(
            self._h, d_nodes, n_stride, d_n_per_scan, B, C.byref(p), d_slot, slot_points, d_cursor,
            d_scan_start, d_n_points, d_status))

    def pack_cloud_xyi_dev(self, d_arena: int, d_cursor: int, slot_points: int, d_slot: int):
        self._check(self._lib.rplgpu_pack_cloud_xyi_dev(self._h, d_arena, d_cursor, slot_points, d_slot))

    def gather_clouds_dev(self, root: int, d_points_local: int, slot_points: int, point_floats: int,
                          d_meta_local: int, meta_words: int, d_points_all: int = 0, d_meta_all: int = 0):
        """Gather to one rank (grouped ncclSend / ncclRecv): the all-gather's layout on `root` only."""
        self._check(self._lib.rplgpu_gather_clouds_dev(
            self._h, root, d_points_local, slot_points, point_floats, d_meta_local, meta_words,
            d_points_all, d_meta_all))

    def allgather_clouds_xyi_dev(self, d_slot_local: int, slot_points: int, d_meta_local: int,
                                 meta_words: int, d_slots_all: int, d_meta_all: int):
        self._check(self._lib.rplgpu_allgather_clouds_xyi_dev(
            self._h, d_slot_local, slot_points, d_meta_local, meta_words, d_slots_all, d_meta_all))

    def unpack_gathered_xyi_dev(self, d_slots_all: int, slot_points: int, d_meta_all: int,
                                meta_words: int, world: int, max_scans: int, d_packed: int,
                                d_total: int, d_scan_start_all: int, d_n_points_all: int,
                                d_status: int = 0):
        self._check(self._lib.rplgpu_unpack_gathered_xyi_dev(
            self._h, d_slots_all, slot_points, d_meta_all, meta_words, world, max_scans, d_packed,
            d_total, d_scan_start_all, d_n_points_all, d_status))

    def comm_fence(self, lag: int = 0):
        self._check(self._lib.rplgpu_comm_fence_lag(self._h, lag))

    def unpack_gathered_dev(self, d_points_all: int, slot_points: int, d_meta_all: int,
                            meta_words: int, world: int, max_scans: int, d_packed: int,
                            d_total: int, d_scan_start_all: int, d_n_points_all: int,
                            d_status: int = 0):
        self._check(self._lib.rplgpu_unpack_gathered_dev(
            self._h, d_points_all, slot_points, d_meta_all, meta_words, world, max_scans, d_packed,
            d_total, d_scan_start_all, d_n_points_all, d_status))

    def fused_cloud_msg_dev(self, d_arena: int, d_total_points: int, arena_capacity: int,
                            frame_id: str, sec: int, nanosec: int, d_msg: int, msg_capacity: int,
                            d_msg_len: int, d_status: int = 0):
        """The whole arena as one serialised PointCloud2 in device memory."""
        self._check(self._lib.rplgpu_fused_cloud_msg_dev(
            self._h, d_arena, d_total_points, arena_capacity, frame_id.encode(),
            Stamp(sec, nanosec), d_msg, msg_capacity, d_msg_len, d_status))

    # -- decode stage: recorded answer streams -> nodes -> scans -----
    def decode_stream(self, ans_type: int, data: np.ndarray, sample_duration_us: int = 125,
                      state=(0, 0)):
        """One stream, host buffers: host framing + GPU decode.  Returns
        ``(nodes, reset_at, n_errors, state_out)``."""
        data = np.ascontiguousarray(data, np.uint8)
        S = self._lib.rplgpu_frame_size(ans_type)
        npf = self._lib.rplgpu_nodes_per_frame(ans_type)
        cap = (len(data) // max(S, 1) + 1) * npf
        nodes = np.zeros(max(cap, 1), NODE_DTYPE)
        rst = np.zeros(len(data) // max(S, 1) + 2, np.uint32)
        st = np.zeros(4, np.int32)
        st[: len(state)] = state
        n, nr, ne = C.c_size_t(0), C.c_size_t(0), C.c_uint32(0)
        self._check(self._lib.rplgpu_decode_stream(
            self._h, ans_type, sample_duration_us, data.ctypes.data, len(data), st.ctypes.data,
            nodes.ctypes.data, len(nodes), C.byref(n), rst.ctypes.data, len(rst), C.byref(nr),
            C.byref(ne)))
        return nodes[: n.value], rst[: nr.value], int(ne.value), (int(st[0]), int(st[1]))

    def decode_batch_dev(self, ans_type: int, sample_duration_us: int, d_bytes: int,
                         stream_stride: int, d_frame_off: int, d_gap: int, d_n_frames: int,
                         max_frames: int, B: int, d_state_in: int, d_state_out: int, d_nodes: int,
                         node_stride: int, d_n_nodes: int, d_reset_at: int = 0,
                         reset_stride: int = 0, d_n_reset: int = 0, d_n_errors: int = 0,
                         d_status: int = 0):
        self._check(self._lib.rplgpu_decode_batch_dev(
            self._h, ans_type, sample_duration_us, d_bytes, stream_stride, d_frame_off, d_gap,
            d_n_frames, max_frames, B, d_state_in, d_state_out, d_nodes, node_stride, d_n_nodes,
            d_reset_at, reset_stride, d_n_reset, d_n_errors, d_status))

    def decode_scans_dev(self, ans_type: int, sample_duration_us: int, d_bytes: int,
                         stream_stride: int, d_frame_off: int, d_gap: int, d_n_frames: int,
                         max_frames: int, B: int, d_state_in: int, d_state_out: int,
                         max_count: int, d_batch: int, n_stride: int, scan_cap: int,
                         d_n_per_scan: int, d_n_scans: int, d_n_errors: int, d_status: int):
        """Recorded streams -> completed scans in batch slots b*scan_cap + s (one call)."""
        self._check(self._lib.rplgpu_decode_scans_dev(
            self._h, ans_type, sample_duration_us, d_bytes, stream_stride, d_frame_off, d_gap,
            d_n_frames, max_frames, B, d_state_in, d_state_out, max_count, d_batch, n_stride,
            scan_cap, d_n_per_scan, d_n_scans, d_n_errors, d_status))

    def decode_scans_carry_dev(self, ans_type: int, sample_duration_us: int, d_bytes: int,
                               stream_stride: int, d_frame_off: int, d_gap: int, d_n_frames: int,
                               max_frames: int, B: int, d_state_in: int, d_state_out: int,
                               max_count: int, d_batch: int, n_stride: int, scan_cap: int,
                               d_n_per_scan: int, d_n_scans: int, d_n_errors: int, d_status: int,
                               d_carry_in: int, d_carry_len_in: int, d_carry_out: int,
                               d_carry_len_out: int, carry_stride: int):
        """decode_scans_dev for one piece of a longer recording: the scan open at the end of the
        call leaves in d_carry_out and enters the next call as d_carry_in."""
        self._check(self._lib.rplgpu_decode_scans_carry_dev(
            self._h, ans_type, sample_duration_us, d_bytes, stream_stride, d_frame_off or None,
            d_gap or None, d_n_frames, max_frames, B, d_state_in or None, d_state_out or None,
            max_count, d_batch, n_stride, scan_cap, d_n_per_scan, d_n_scans, d_n_errors or None,
            d_status, d_carry_in or None, d_carry_len_in or None, d_carry_out, d_carry_len_out,
            carry_stride))

    def segment_batch_dev(self, d_nodes: int, node_stride: int, d_n_nodes: int, d_reset_at: int,
                          reset_stride: int, d_n_reset: int, B: int, max_count: int,
                          d_out_nodes: int, out_stride: int, d_scan_off: int, scan_cap: int,
                          d_n_scans: int, d_status: int = 0):
        self._check(self._lib.rplgpu_segment_batch_dev(
            self._h, d_nodes, node_stride, d_n_nodes, d_reset_at, reset_stride, d_n_reset, B,
            max_count, d_out_nodes, out_stride, d_scan_off, scan_cap, d_n_scans, d_status))

    def scans_to_batch_dev(self, d_seg_nodes: int, seg_stride: int, d_scan_off: int,
                           scan_cap: int, d_n_scans: int, B: int, d_scan_base: int, d_batch: int,
                           n_stride: int, max_scans: int, d_n_per_scan: int):
        self._check(self._lib.rplgpu_scans_to_batch_dev(
            self._h, d_seg_nodes, seg_stride, d_scan_off, scan_cap, d_n_scans, B, d_scan_base,
            d_batch, n_stride, max_scans, d_n_per_scan))


def cloud_meta_words(max_scans: int) -> int:
    return int(load_library().rplgpu_cloud_meta_words(max_scans))


def msg_laserscan_layout(frame_id_len: int, count: int) -> LaserScanLayout:
    L = LaserScanLayout()
    rc = load_library().rplgpu_msg_laserscan_layout(frame_id_len, count, C.byref(L))
    if rc:
        raise RplGpuError(rc, "rplgpu_msg_laserscan_layout")
    return L


def msg_cloud_layout(frame_id_len: int, n_points: int) -> CloudLayout:
    L = CloudLayout()
    rc = load_library().rplgpu_msg_cloud_layout(frame_id_len, n_points, C.byref(L))
    if rc:
        raise RplGpuError(rc, "rplgpu_msg_cloud_layout")
    return L


def msg_laserscan_header(frame_id: str, sec: int, nanosec: int, meta: ScanMeta,
                         out: np.ndarray) -> LaserScanLayout:
    """Host-only: everything of a serialised LaserScan but the two float arrays, into ``out``."""
    L = LaserScanLayout()
    rc = load_library().rplgpu_msg_laserscan_header(
        frame_id.encode(), Stamp(sec, nanosec), C.byref(meta), out.ctypes.data, out.nbytes,
        C.byref(L))
    if rc:
        raise RplGpuError(rc, "rplgpu_msg_laserscan_header")
    return L


def msg_cloud_header(frame_id: str, sec: int, nanosec: int, n_points: int,
                     out: np.ndarray) -> CloudLayout:
    """Host-only: everything of a serialised PointCloud2 but the points, into ``out``."""
    L = CloudLayout()
    rc = load_library().rplgpu_msg_cloud_header(
        frame_id.encode(), Stamp(sec, nanosec), n_points, out.ctypes.data, out.nbytes, C.byref(L))
    if rc:
        raise RplGpuError(rc, "rplgpu_msg_cloud_header")
    return L


def frame_stream(ans_type: int, data: np.ndarray):
    """Host framing (``rplgpu_frame_stream``): ``(frame_off, gap)`` of a recorded byte stream."""
    lib = load_library()
    data = np.ascontiguousarray(data, np.uint8)
    S = lib.rplgpu_frame_size(ans_type)
    if not S:
        raise ValueError(f"unknown answer type {ans_type:#x}")
    cap = len(data) // S + 1
    off = np.zeros(cap, np.uint32)
    gap = np.zeros(cap, np.uint8)
    nf = lib.rplgpu_frame_stream(ans_type, data.ctypes.data, len(data), off.ctypes.data,
                                 gap.ctypes.data, cap)
    return off[:nf], gap[:nf]


# -- host twins of the exchange layout (include/rplgpu_comm.h; no device, no handle) -----------
def pack_cloud_meta_host(cursor: int, scan_start: np.ndarray, n_points: np.ndarray, slot_points: int,
                         max_scans: int) -> np.ndarray:
    """META block of a rank's arena, as ``rplgpu_pack_cloud_meta_dev`` writes it."""
    lib = load_library()
    ss = np.ascontiguousarray(scan_start, np.uint64)
    npnt = np.ascontiguousarray(n_points, np.uint32)
    words = int(lib.rplgpu_cloud_meta_words(max_scans))
    meta = np.zeros(words, np.uint32)
    rc = lib.rplgpu_pack_cloud_meta_host(int(cursor), ss.ctypes.data, npnt.ctypes.data, len(npnt),
                                         int(slot_points), int(max_scans), meta.ctypes.data)
    if rc:
        raise RplGpuError(rc, "rplgpu_pack_cloud_meta_host")
    return meta


def pack_cloud_xyi_host(arena: np.ndarray, cursor: int, slot_points: int) -> np.ndarray:
    """A rank's compact slot (slot_points x 3 float32) from its arena ((cap, 4) float32)."""
    a = np.ascontiguousarray(arena, np.float32)
    slot = np.zeros((int(slot_points), 3), np.float32)
    rc = load_library().rplgpu_pack_cloud_xyi_host(a.ctypes.data, int(cursor), int(slot_points),
                                                   slot.ctypes.data)
    if rc:
        raise RplGpuError(rc, "rplgpu_pack_cloud_xyi_host")
    return slot


def unpack_gathered_host(points_all: np.ndarray, slot_points: int, meta_all: np.ndarray, world: int,
                         max_scans: int):
    """Gathered slots ((world, slot_points, 3 | 4) float32) + META blocks ((world, words) uint32) ->
    (packed (total, 4), scan_start_all (world, max_scans) uint64, n_points_all (world, max_scans)
    uint32, status (world,) uint32), by the library's own layout code."""
    lib = load_library()
    pts = np.ascontiguousarray(points_all, np.float32)
    point_floats = int(pts.shape[-1])
    meta = np.ascontiguousarray(meta_all, np.uint32).reshape(world, -1)
    packed = np.zeros((world * int(slot_points), 4), np.float32)
    total = np.zeros(1, np.uint64)
    starts = np.zeros((world, max_scans), np.uint64)
    npts = np.zeros((world, max_scans), np.uint32)
    status = np.zeros(world, np.uint32)
    rc = lib.rplgpu_unpack_gathered_host(pts.ctypes.data, int(slot_points), point_floats,
                                         meta.ctypes.data, int(meta.shape[1]), int(world),
                                         int(max_scans), packed.ctypes.data, total.ctypes.data,
                                         starts.ctypes.data, npts.ctypes.data, status.ctypes.data)
    if rc:
        raise RplGpuError(rc, "rplgpu_unpack_gathered_host")
    return packed[: int(total[0])], starts, npts, status
