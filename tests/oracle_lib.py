"""ctypes access to the CPU oracle (oracle/liboracle.so) and, when built, to the genuine
reference libraries under oracle/_ref/.  TEST INFRASTRUCTURE — imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
ORACLE_DIR = ROOT / "oracle"

NODE = np.dtype([("angle_z_q14", "<u2"), ("dist_mm_q2", "<u4"), ("quality", "u1"), ("flag", "u1")])


class OParams(C.Structure):
    _fields_ = [
        ("is_new_protocol", C.c_int32), ("inverted", C.c_int32), ("scan_processing", C.c_int32),
        ("clip_enable", C.c_int32), ("q_min", C.c_uint32), ("range_min", C.c_float),
        ("range_max", C.c_float), ("voxel_leaf", C.c_float), ("ror_radius", C.c_float),
        ("ror_min_neighbors", C.c_uint32), ("ror_enable", C.c_int32), ("voxel_enable", C.c_int32),
    ]


class OMeta(C.Structure):
    _fields_ = [(k, C.c_float) for k in
                "angle_min angle_max angle_increment time_increment scan_time range_min range_max".split()
                ] + [("count", C.c_uint32), ("published", C.c_int32)]

    def as_tuple(self):
        return tuple(getattr(self, f[0]) for f in self._fields_)


def params(**kw) -> OParams:
    p = OParams(0, 0, 1, 0, 0, 0.15, 12.0, 0.05, 0.10, 2, 0, 0)
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def copy_params(src) -> OParams:
    """Field-by-field copy from the product's Params structure (same names)."""
    return params(**{f[0]: getattr(src, f[0]) for f in OParams._fields_})


class Oracle:
    def __init__(self, lib: C.CDLL):
        self.lib = lib
        vp, sz = C.c_void_p, C.c_size_t
        lib.orc_ascend.argtypes = [vp, sz]
        lib.orc_ascend.restype = C.c_uint32
        lib.orc_publish_scan.argtypes = [vp, sz, C.POINTER(OParams), C.c_double, vp, vp, C.POINTER(OMeta)]
        lib.orc_publish_scan.restype = None
        lib.orc_effective_max_range.argtypes = [C.c_float, C.c_float]
        lib.orc_effective_max_range.restype = C.c_float
        lib.orc_gen_dummy.argtypes = [C.c_uint32, vp]
        lib.orc_gen_dummy.restype = None
        lib.orc_scan_to_cloud.argtypes = [vp, sz, C.POINTER(OParams), vp]
        lib.orc_scan_to_cloud.restype = sz
        lib.orc_ror_mask.argtypes = [vp, sz, C.c_float, C.c_uint32, vp]
        lib.orc_ror_mask.restype = None
        lib.orc_voxel_grid.argtypes = [vp, sz, C.c_float, vp, vp, vp]
        lib.orc_voxel_grid.restype = sz
        lib.orc_cloud_pipeline.argtypes = [vp, sz, C.POINTER(OParams), vp, vp, vp]
        lib.orc_cloud_pipeline.restype = sz
        lib.orc_laserscan_to_cloud.argtypes = [vp, vp, C.c_uint32, C.POINTER(OParams), vp]
        lib.orc_laserscan_to_cloud.restype = sz
        for name in ("orc_batch_ascend",):
            getattr(lib, name).argtypes = [vp, sz, vp, sz, C.c_int]
            getattr(lib, name).restype = C.c_uint64
        for name in ("orc_batch_laserscan", "orc_batch_cloud"):
            getattr(lib, name).argtypes = [vp, sz, vp, sz, C.POINTER(OParams), C.c_int]
            getattr(lib, name).restype = C.c_uint64
        lib.orc_batch_cloud_check.argtypes = [vp, sz, vp, sz, C.POINTER(OParams), vp, vp, vp, vp,
                                              C.c_int, vp]
        lib.orc_batch_cloud_check.restype = C.c_uint64
        lib.orc_batch_ascend_check.argtypes = [vp, vp, sz, vp, sz, C.c_int, vp]
        lib.orc_batch_ascend_check.restype = C.c_uint64
        lib.orc_batch_laserscan_check.argtypes = [vp, sz, vp, sz, C.POINTER(OParams), vp, vp, vp, sz,
                                                  C.c_int, vp]
        lib.orc_batch_laserscan_check.restype = C.c_uint64

    def batch_ascend_check(self, src: np.ndarray, got: np.ndarray, lens: np.ndarray, threads: int):
        """Every scan of `src` (B x n_stride nodes, `lens[b]` used) through orc_ascend on all host
        threads, compared with the device's in-place result `got`.  Returns (scans with a mismatch,
        per-scan table: sl_result, differing angle words, differing nodes after canonicalising
        equal-angle runs, valid nodes out of stable order)."""
        B, n = src.shape
        src = np.ascontiguousarray(src)
        got = np.ascontiguousarray(got)
        assert got.shape == src.shape and got.dtype == src.dtype
        lens = np.ascontiguousarray(lens, np.uint32)
        res = np.zeros((B, 4), np.uint32)
        bad = int(self.lib.orc_batch_ascend_check(src.ctypes.data, got.ctypes.data, n, lens.ctypes.data, B,
                                                  int(threads), res.ctypes.data))
        return bad, res

    def batch_laserscan_check(self, batch: np.ndarray, lens: np.ndarray, p: "OParams", ranges: np.ndarray,
                              intens: np.ndarray, count: np.ndarray, threads: int):
        """Every scan through orc_publish_scan on all host threads, compared with the device's
        ranges / intensities (B x out_stride) and beam counts.  Returns (scans with a mismatch,
        per-scan table: oracle count, count differs, differing range words, differing intensity
        words not explained by an equal-(angle, dist) tie)."""
        B, n = batch.shape
        batch = np.ascontiguousarray(batch)
        lens = np.ascontiguousarray(lens, np.uint32)
        ranges = np.ascontiguousarray(ranges, np.float32)
        intens = np.ascontiguousarray(intens, np.float32)
        count = np.ascontiguousarray(count, np.uint32)
        assert ranges.shape == intens.shape and ranges.shape[0] == B
        res = np.zeros((B, 4), np.uint32)
        bad = int(self.lib.orc_batch_laserscan_check(
            batch.ctypes.data, n, lens.ctypes.data, B, C.byref(p), ranges.ctypes.data, intens.ctypes.data,
            count.ctypes.data, ranges.shape[1], int(threads), res.ctypes.data))
        return bad, res

    def batch_cloud_check(self, batch: np.ndarray, p: "OParams", arena: np.ndarray, start: np.ndarray,
                          npts: np.ndarray, keys, threads: int):
        """Every scan of `batch` through the cloud oracle (all host threads), compared with the
        device's output.  Returns (scans with a count / key / intensity mismatch, per-scan table:
        oracle cells, bad keys, bad intensities, max |dx|,|dy|)."""
        B, n = batch.shape
        nodes = np.ascontiguousarray(batch)
        lens = np.full(B, n, np.uint32)
        arena = np.ascontiguousarray(arena, np.float32)
        start = np.ascontiguousarray(start, np.uint64)
        npts = np.ascontiguousarray(npts, np.uint32)
        keys = None if keys is None else np.ascontiguousarray(keys, np.uint32)
        res = np.zeros((B, 4), np.uint32)
        bad = int(self.lib.orc_batch_cloud_check(
            nodes.ctypes.data, n, lens.ctypes.data, B, C.byref(p), arena.ctypes.data,
            start.ctypes.data, npts.ctypes.data, None if keys is None else keys.ctypes.data,
            int(threads), res.ctypes.data))
        return bad, res

    def ascend(self, nodes: np.ndarray):
        out = np.ascontiguousarray(nodes).copy()
        res = self.lib.orc_ascend(out.ctypes.data, len(out))
        return out, res

    # ---- unpackers + scan assembly (oracle_unpack.cpp) ------------------
    def _bind_unpack(self):
        lib, vp, sz = self.lib, C.c_void_p, C.c_size_t
        if getattr(lib, "_unpack_bound", False):
            return
        lib.orc_frame_size.argtypes = [C.c_uint8]
        lib.orc_frame_size.restype = sz
        lib.orc_frame_stream.argtypes = [C.c_uint8, vp, sz, vp, vp, sz]
        lib.orc_frame_stream.restype = sz
        lib.orc_unpack_frames.argtypes = [C.c_uint8, vp, vp, vp, sz, C.c_uint32, vp, vp, sz, vp,
                                          sz, C.POINTER(sz), C.POINTER(C.c_uint32)]
        lib.orc_unpack_frames.restype = sz
        lib.orc_unpack.argtypes = [C.c_uint8, vp, sz, C.c_uint32, vp, vp, sz, vp, sz,
                                   C.POINTER(sz), C.POINTER(C.c_uint32)]
        lib.orc_unpack.restype = sz
        lib.orc_segment.argtypes = [vp, sz, vp, sz, sz, vp, sz, vp, sz]
        lib.orc_segment.restype = sz
        lib._unpack_bound = True

    def frame_stream(self, ans: int, data: np.ndarray):
        self._bind_unpack()
        data = np.ascontiguousarray(data, np.uint8)
        S = self.lib.orc_frame_size(ans)
        cap = len(data) // S + 1
        off = np.zeros(cap, np.uint32)
        gap = np.zeros(cap, np.uint8)
        nf = self.lib.orc_frame_stream(ans, data.ctypes.data, len(data), off.ctypes.data,
                                       gap.ctypes.data, cap)
        return off[:nf], gap[:nf]

    def unpack_frames(self, ans: int, data, off, gap, sample_duration_us: int = 125,
                      state=(0, 0)):
        self._bind_unpack()
        data = np.ascontiguousarray(data, np.uint8)
        off = np.ascontiguousarray(off, np.uint32)
        gap = np.ascontiguousarray(gap, np.uint8)
        cap = len(off) * 96 + 1
        out = np.zeros(cap, NODE)
        rst = np.zeros(len(off) + 1, np.uint32)
        st = np.array(state, np.int32)
        nr, ne = C.c_size_t(0), C.c_uint32(0)
        n = self.lib.orc_unpack_frames(ans, data.ctypes.data, off.ctypes.data, gap.ctypes.data,
                                       len(off), sample_duration_us, st.ctypes.data,
                                       out.ctypes.data, cap, rst.ctypes.data, len(rst),
                                       C.byref(nr), C.byref(ne))
        return out[:n], rst[: nr.value], int(ne.value), (int(st[0]), int(st[1]))

    def unpack(self, ans: int, data, sample_duration_us: int = 125, state=(0, 0)):
        self._bind_unpack()
        data = np.ascontiguousarray(data, np.uint8)
        S = self.lib.orc_frame_size(ans)
        cap = (len(data) // S + 1) * 96 + 1
        out = np.zeros(cap, NODE)
        rst = np.zeros(len(data) // S + 2, np.uint32)
        st = np.array(state, np.int32)
        nr, ne = C.c_size_t(0), C.c_uint32(0)
        n = self.lib.orc_unpack(ans, data.ctypes.data, len(data), sample_duration_us,
                                st.ctypes.data, out.ctypes.data, cap, rst.ctypes.data, len(rst),
                                C.byref(nr), C.byref(ne))
        return out[:n], rst[: nr.value], int(ne.value), (int(st[0]), int(st[1]))

    def segment(self, nodes, reset_at, max_count: int = 8192):
        self._bind_unpack()
        nodes = np.ascontiguousarray(nodes)
        reset_at = np.ascontiguousarray(reset_at, np.uint32)
        out = np.zeros(max(len(nodes), 1), NODE)
        offs = np.zeros(len(nodes) + 2, np.uint32)
        ns = self.lib.orc_segment(nodes.ctypes.data, len(nodes), reset_at.ctypes.data,
                                  len(reset_at), max_count, out.ctypes.data, len(out),
                                  offs.ctypes.data, len(offs))
        return out[: offs[ns]], offs[: ns + 1]

    def publish_scan(self, nodes: np.ndarray, p: OParams, scan_duration: float = 0.1):
        nodes = np.ascontiguousarray(nodes)
        n = len(nodes)
        r = np.full(max(n, 1), np.nan, np.float32)
        i = np.full(max(n, 1), np.nan, np.float32)
        m = OMeta()
        self.lib.orc_publish_scan(nodes.ctypes.data, n, C.byref(p), scan_duration,
                                  r.ctypes.data, i.ctypes.data, C.byref(m))
        return r[: m.count], i[: m.count], m

    def gen_dummy(self, scan_index: int = 0) -> np.ndarray:
        out = np.zeros(360, NODE)
        self.lib.orc_gen_dummy(scan_index, out.ctypes.data)
        return out

    def scan_to_cloud(self, nodes: np.ndarray, p: OParams) -> np.ndarray:
        nodes = np.ascontiguousarray(nodes)
        out = np.zeros((max(len(nodes), 1), 4), np.float32)
        m = self.lib.orc_scan_to_cloud(nodes.ctypes.data, len(nodes), C.byref(p), out.ctypes.data)
        return out[:m]

    def laserscan_to_cloud(self, ranges: np.ndarray, intens: np.ndarray, p: OParams) -> np.ndarray:
        ranges = np.ascontiguousarray(ranges, np.float32)
        intens = np.ascontiguousarray(intens, np.float32)
        out = np.zeros((max(len(ranges), 1), 4), np.float32)
        m = self.lib.orc_laserscan_to_cloud(ranges.ctypes.data, intens.ctypes.data, len(ranges),
                                            C.byref(p), out.ctypes.data)
        return out[:m]

    def voxel_grid(self, xyzi: np.ndarray, leaf: float):
        xyzi = np.ascontiguousarray(xyzi, np.float32)
        n = len(xyzi)
        out = np.zeros((max(n, 1), 4), np.float32)
        cells = np.zeros((max(n, 1), 2), np.int32)
        counts = np.zeros(max(n, 1), np.uint32)
        m = self.lib.orc_voxel_grid(xyzi.ctypes.data, n, leaf, out.ctypes.data,
                                    cells.ctypes.data, counts.ctypes.data)
        return out[:m], cells[:m], counts[:m]

    def cloud_pipeline(self, nodes: np.ndarray, p: OParams):
        nodes = np.ascontiguousarray(nodes)
        n = len(nodes)
        out = np.zeros((max(n, 1), 4), np.float32)
        cells = np.zeros((max(n, 1), 2), np.int32)
        counts = np.zeros(max(n, 1), np.uint32)
        m = self.lib.orc_cloud_pipeline(nodes.ctypes.data, n, C.byref(p), out.ctypes.data,
                                        cells.ctypes.data, counts.ctypes.data)
        return out[:m], cells[:m], counts[:m]

    def ror_mask(self, xyzi: np.ndarray, radius: float, k: int) -> np.ndarray:
        xyzi = np.ascontiguousarray(xyzi, np.float32)
        keep = np.zeros(max(len(xyzi), 1), np.uint8)
        self.lib.orc_ror_mask(xyzi.ctypes.data, len(xyzi), radius, k, keep.ctypes.data)
        return keep[: len(xyzi)].astype(bool)


def build_oracle():
    subprocess.run(["make", "-C", str(ORACLE_DIR), "all"], check=True, capture_output=True)


def load_oracle() -> Oracle:
    so = ORACLE_DIR / "liboracle.so"
    src_newer = (not so.exists()) or any(
        (ORACLE_DIR / f).stat().st_mtime > so.stat().st_mtime
        for f in ("oracle.cpp", "oracle_unpack.cpp", "oracle.h"))
    if src_newer:
        build_oracle()
    return Oracle(C.CDLL(str(so)))


class RefLibs:
    """Genuine reference code: SDK ascendScanData and RPlidarNode::publish_scan."""

    def __init__(self, sl: C.CDLL, node: C.CDLL):
        self.sl, self.node = sl, node
        sl.ref_ascend.argtypes = [C.c_void_p, C.c_size_t]
        sl.ref_ascend.restype = C.c_uint32
        node.ref_publish_scan.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                          C.c_float, C.c_double, C.c_void_p, C.c_void_p,
                                          C.POINTER(OMeta)]
        node.ref_dummy_grab.argtypes = [C.c_void_p, C.c_size_t]

    def ascend(self, nodes: np.ndarray):
        out = np.ascontiguousarray(nodes).copy()
        res = self.sl.ref_ascend(out.ctypes.data, len(out))
        return out, res

    def publish_scan(self, nodes, *, driver_kind: int, inverted: int, scan_processing: int,
                     range_max: float = 12.0, scan_duration: float = 0.1):
        nodes = np.ascontiguousarray(nodes)
        n = len(nodes)
        r = np.full(max(n, 1), np.nan, np.float32)
        i = np.full(max(n, 1), np.nan, np.float32)
        m = OMeta()
        self.node.ref_publish_scan(nodes.ctypes.data, n, driver_kind, inverted, scan_processing,
                                   range_max, scan_duration, r.ctypes.data, i.ctypes.data,
                                   C.byref(m))
        return r[: m.count], i[: m.count], m

    def dummy_grab(self) -> np.ndarray:
        out = np.zeros(360, NODE)
        k = self.node.ref_dummy_grab(out.ctypes.data, 360)
        assert k == 360
        return out


def load_ref():
    a, b = ORACLE_DIR / "_ref" / "libslref.so", ORACLE_DIR / "_ref" / "libnoderef.so"
    if not (a.exists() and b.exists()):
        return None
    return RefLibs(C.CDLL(str(a)), C.CDLL(str(b)))


class RefUnpack:
    """Genuine reference unpackers + ScanDataHolder (oracle/_ref/libunpackref.so)."""

    def __init__(self, lib: C.CDLL):
        self.lib = lib
        vp, sz = C.c_void_p, C.c_size_t
        lib.ref_unpack.argtypes = [C.c_uint8, vp, sz, sz, C.c_uint32, vp, sz, vp, sz,
                                   C.POINTER(sz), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        lib.ref_unpack.restype = sz
        lib.ref_segment.argtypes = [vp, sz, vp, sz, sz, vp, sz, vp, sz]
        lib.ref_segment.restype = sz

    def unpack(self, ans: int, data, sample_duration_us: int = 125, chunk: int = 0):
        data = np.ascontiguousarray(data, np.uint8)
        cap = (len(data) // 4 + 1) * 4 + 200  # >= nodes any answer type can publish
        out = np.zeros(cap, NODE)
        rst = np.zeros(len(data) // 5 + 2, np.uint32)
        nr, ne, nenc = C.c_size_t(0), C.c_uint32(0), C.c_uint32(0)
        n = self.lib.ref_unpack(ans, data.ctypes.data, len(data), chunk, sample_duration_us,
                                out.ctypes.data, cap, rst.ctypes.data, len(rst), C.byref(nr),
                                C.byref(ne), C.byref(nenc))
        assert n <= cap
        return out[:n], rst[: nr.value], int(ne.value)

    def segment(self, nodes, reset_at, max_count: int = 8192):
        nodes = np.ascontiguousarray(nodes)
        reset_at = np.ascontiguousarray(reset_at, np.uint32)
        out = np.zeros(max(len(nodes), 1), NODE)
        offs = np.zeros(len(nodes) + 2, np.uint32)
        ns = self.lib.ref_segment(nodes.ctypes.data, len(nodes), reset_at.ctypes.data,
                                  len(reset_at), max_count, out.ctypes.data, len(out),
                                  offs.ctypes.data, len(offs))
        return out[: offs[ns]], offs[: ns + 1]


def load_ref_unpack():
    so = ORACLE_DIR / "_ref" / "libunpackref.so"
    return RefUnpack(C.CDLL(str(so))) if so.exists() else None


def canon_equal_angle_runs(nodes: np.ndarray) -> np.ndarray:
    """Canonical form for comparing ascend outputs: inside every run of equal angle the
    reference order is whatever introsort leaves (unstable), so sort such runs by the
    remaining fields.  Angles themselves must already be ascending."""
    key = np.lexsort((nodes["flag"], nodes["quality"], nodes["dist_mm_q2"], nodes["angle_z_q14"]))
    srt = nodes[key]
    return srt
