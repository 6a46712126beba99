"""CDR restatement for the serialised-message stage.

TEST INFRASTRUCTURE ONLY — imported by tests/ and nothing else; the product
(include/rplgpu_msg.h) never calls it.

PARITY UNPINNED for the wire format: the reference publishes typed messages
(`scan_pub_->publish(scan_msg)`, src/rplidar_node.cpp:682) and leaves serialisation to the
middleware — rosidl_typesupport_fastrtps_cpp + eProsima Fast-CDR (ROS 2 Jazzy per the
reference README: Fast-CDR 2.2.x, XCDR version 1, PLAIN_CDR, little endian), none of which is
in /root/reference or in this image.  This file restates the published rules (OMG
DDS-XTypes 1.3 §7.4.1, as Fast-CDR applies them to ROS 2 messages) generically: a message is
described by its .msg definition (field list) and serialised / parsed by walking that list —
deliberately a different construction from the product's fixed-function writer.

Rules:
  * 4-byte encapsulation header 00 01 00 00, then the body;
  * a primitive of size s starts at a multiple of s, counted from the first body byte;
  * string: uint32 length INCLUDING the terminating NUL, the bytes, the NUL;
  * T[] (unbounded sequence): uint32 element count, then the elements (each aligned as T);
  * nested message: its fields in order, no extra alignment;
  * nothing follows the last member.

Message definitions (ROS 2 common_interfaces, unchanged since Foxy):
  builtin_interfaces/Time   int32 sec, uint32 nanosec
  std_msgs/Header           Time stamp, string frame_id
  sensor_msgs/LaserScan     Header header, float32 angle_min, angle_max, angle_increment,
                            time_increment, scan_time, range_min, range_max,
                            float32[] ranges, float32[] intensities
  sensor_msgs/PointField    string name, uint32 offset, uint8 datatype, uint32 count
  sensor_msgs/PointCloud2   Header header, uint32 height, uint32 width, PointField[] fields,
                            bool is_bigendian, uint32 point_step, uint32 row_step,
                            uint8[] data, bool is_dense
"""
from __future__ import annotations

import struct

import numpy as np

PRIM = {  # name -> (struct code, size)
    "bool": ("?", 1), "uint8": ("B", 1), "int32": ("i", 4), "uint32": ("I", 4),
    "float32": ("f", 4), "float64": ("d", 8),
}

MSGS = {
    "Time": [("sec", "int32"), ("nanosec", "uint32")],
    "Header": [("stamp", "Time"), ("frame_id", "string")],
    "LaserScan": [("header", "Header"), ("angle_min", "float32"), ("angle_max", "float32"),
                  ("angle_increment", "float32"), ("time_increment", "float32"),
                  ("scan_time", "float32"), ("range_min", "float32"), ("range_max", "float32"),
                  ("ranges", "float32[]"), ("intensities", "float32[]")],
    "PointField": [("name", "string"), ("offset", "uint32"), ("datatype", "uint8"),
                   ("count", "uint32")],
    "PointCloud2": [("header", "Header"), ("height", "uint32"), ("width", "uint32"),
                    ("fields", "PointField[]"), ("is_bigendian", "bool"),
                    ("point_step", "uint32"), ("row_step", "uint32"), ("data", "uint8[]"),
                    ("is_dense", "bool")],
}

ENCAPSULATION = bytes([0x00, 0x01, 0x00, 0x00])
FLOAT32 = 7  # sensor_msgs/PointField.FLOAT32


class _Out:
    def __init__(self):
        self.body = bytearray()

    def align(self, size):
        while len(self.body) % size:
            self.body.append(0)

    def prim(self, code, size, value):
        self.align(size)
        self.body += struct.pack("<" + code, value)


def _ser(out: _Out, typ: str, value):
    if typ.endswith("[]"):
        elem = typ[:-2]
        out.prim("I", 4, len(value))
        if elem in PRIM and not isinstance(value, (list, tuple)):
            arr = np.ascontiguousarray(value)
            if len(arr):
                out.align(PRIM[elem][1])
            out.body += arr.tobytes()
        else:
            for v in value:
                _ser(out, elem, v)
    elif typ == "string":
        raw = value.encode() if isinstance(value, str) else bytes(value)
        out.prim("I", 4, len(raw) + 1)
        out.body += raw + b"\0"
    elif typ in PRIM:
        out.prim(*PRIM[typ], value)
    else:
        for name, ftyp in MSGS[typ]:
            _ser(out, ftyp, value[name])


def serialize(typ: str, value: dict) -> bytes:
    out = _Out()
    _ser(out, typ, value)
    return ENCAPSULATION + bytes(out.body)


class _In:
    def __init__(self, body: bytes):
        self.body = body
        self.pos = 0

    def prim(self, code, size):
        self.pos += (-self.pos) % size
        (v,) = struct.unpack_from("<" + code, self.body, self.pos)
        self.pos += size
        return v


def _de(inp: _In, typ: str):
    if typ.endswith("[]"):
        elem = typ[:-2]
        n = inp.prim("I", 4)
        if elem in PRIM:
            code, size = PRIM[elem]
            if n:
                inp.pos += (-inp.pos) % size
            dt = {"f": "<f4", "B": "u1", "I": "<u4", "i": "<i4", "d": "<f8", "?": "u1"}[code]
            arr = np.frombuffer(inp.body, dtype=dt, count=n, offset=inp.pos).copy()
            inp.pos += n * size
            return arr
        return [_de(inp, elem) for _ in range(n)]
    if typ == "string":
        n = inp.prim("I", 4)
        raw = inp.body[inp.pos: inp.pos + n]
        inp.pos += n
        assert n >= 1 and raw[-1] == 0, "string without terminating NUL"
        return raw[:-1].decode()
    if typ in PRIM:
        return inp.prim(*PRIM[typ])
    return {name: _de(inp, ftyp) for name, ftyp in MSGS[typ]}


def deserialize(typ: str, msg: bytes) -> dict:
    assert bytes(msg[:4]) == ENCAPSULATION, "not little-endian plain CDR"
    inp = _In(bytes(msg[4:]))
    val = _de(inp, typ)
    assert inp.pos == len(inp.body), f"{len(inp.body) - inp.pos} trailing bytes"
    return val


def laserscan_msg(frame_id: str, sec: int, nanosec: int, meta, ranges, intensities) -> bytes:
    """meta: object/dict with angle_min .. range_max (the node's scalars, :618-627)."""
    g = (lambda k: meta[k]) if isinstance(meta, dict) else (lambda k: getattr(meta, k))
    return serialize("LaserScan", {
        "header": {"stamp": {"sec": sec, "nanosec": nanosec}, "frame_id": frame_id},
        "angle_min": g("angle_min"), "angle_max": g("angle_max"),
        "angle_increment": g("angle_increment"), "time_increment": g("time_increment"),
        "scan_time": g("scan_time"), "range_min": g("range_min"), "range_max": g("range_max"),
        "ranges": np.asarray(ranges, dtype="<f4"),
        "intensities": np.asarray(intensities, dtype="<f4"),
    })


def cloud_msg(frame_id: str, sec: int, nanosec: int, xyzi) -> bytes:
    """xyzi: (n, 4) float32 — the E3 layout of SURVEY.md §8(a)."""
    pts = np.ascontiguousarray(np.asarray(xyzi, dtype="<f4").reshape(-1, 4))
    n = pts.shape[0]
    fields = [{"name": nm, "offset": off, "datatype": FLOAT32, "count": 1}
              for nm, off in (("x", 0), ("y", 4), ("z", 8), ("intensity", 12))]
    return serialize("PointCloud2", {
        "header": {"stamp": {"sec": sec, "nanosec": nanosec}, "frame_id": frame_id},
        "height": 1, "width": n, "fields": fields, "is_bigendian": False,
        "point_step": 16, "row_step": 16 * n,
        "data": pts.view(np.uint8).reshape(-1), "is_dense": True,
    })
