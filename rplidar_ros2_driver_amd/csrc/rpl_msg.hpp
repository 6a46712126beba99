// rpl_msg.hpp — serialised (CDR) LaserScan / PointCloud2 framing shared by the host writers
// and the device assembly kernels (include/rplgpu_msg.h states the wire format).
//
// A message is: a PREFIX (encapsulation header, std_msgs/Header, scalars, field table) whose
// bytes are the same for every scan of a publisher except a handful of patched words, then
// the bulk arrays, then (PointCloud2) one trailing byte.  The host builds the prefix once per
// call (`Prefix`), the kernels patch and copy it.
#pragma once
#include <stdint.h>
#include <string.h>

#include "rplgpu_msg.h"

namespace rplmsg {

constexpr uint32_t kMaxFrameId = 255;    // strlen(frame_id) the device template has room for
constexpr uint32_t kPrefixWords = 112;   // 448 bytes >= PointCloud2 prefix with a 255-byte id

// Prefix bytes + where the per-scan words go (byte offsets from the start of the message;
// every one a multiple of 4).  Passed to the kernels by value.
struct Prefix {
  uint32_t len;        // prefix length in bytes (multiple of 4) == offset of the first array
  uint32_t stamp_off;  // int32 sec, uint32 nanosec
  uint32_t a_off;      // LaserScan: angle_min (7 scalars follow) | PointCloud2: width
  uint32_t b_off;      // LaserScan: ranges length word         | PointCloud2: row_step
  uint32_t c_off;      // LaserScan: unused                      | PointCloud2: data length word
  uint32_t words[kPrefixWords];
};

// Sequential little-endian CDR writer; p == nullptr only measures.  Alignment is counted from
// the byte after the 4-byte encapsulation header.
class Writer {
 public:
  Writer(uint8_t *p, size_t cap) : p_(p), cap_(cap) {}
  size_t pos() const { return pos_; }
  void u8(uint8_t v) { put(&v, 1); }
  void u32(uint32_t v) { align4(); put(&v, 4); }
  void i32(int32_t v) { align4(); put(&v, 4); }
  void f32(float v) { align4(); put(&v, 4); }
  void str(const char *s, size_t n) {  // uint32 length incl. NUL, bytes, NUL
    u32(static_cast<uint32_t>(n + 1));
    put(s, n);
    u8(0);
  }
  void align4() {
    while ((pos_ - 4) & 3u) u8(0);
  }
  void skip(size_t n) { pos_ += n; }  // bulk array filled by someone else
 private:
  void put(const void *src, size_t n) {
    if (p_ && pos_ + n <= cap_) memcpy(p_ + pos_, src, n);
    pos_ += n;
  }
  uint8_t *p_;
  size_t cap_;
  size_t pos_ = 0;
};

inline void encapsulation(Writer &w) {
  w.u8(0x00), w.u8(0x01), w.u8(0x00), w.u8(0x00);  // CDR_LE, options 0
}

inline void header(Writer &w, const char *frame_id, size_t fid_len, rplgpu_stamp_t stamp) {
  w.i32(stamp.sec);
  w.u32(stamp.nanosec);
  w.str(frame_id, fid_len);
}

// Whole LaserScan but the two float arrays.  Returns total length.
inline size_t write_laserscan(Writer &w, const char *frame_id, size_t fid_len,
                              rplgpu_stamp_t stamp, const rplgpu_scan_meta_t &m,
                              rplgpu_laserscan_layout_t *L) {
  encapsulation(w);
  header(w, frame_id, fid_len, stamp);
  w.align4();
  L->scalars_off = static_cast<uint32_t>(w.pos());
  w.f32(m.angle_min), w.f32(m.angle_max), w.f32(m.angle_increment), w.f32(m.time_increment);
  w.f32(m.scan_time), w.f32(m.range_min), w.f32(m.range_max);
  L->ranges_len_off = static_cast<uint32_t>(w.pos());
  w.u32(m.count);
  L->ranges_off = static_cast<uint32_t>(w.pos());
  w.skip(static_cast<size_t>(m.count) * 4);
  L->intensities_len_off = static_cast<uint32_t>(w.pos());
  w.u32(m.count);
  L->intensities_off = static_cast<uint32_t>(w.pos());
  w.skip(static_cast<size_t>(m.count) * 4);
  L->total_len = static_cast<uint32_t>(w.pos());
  return w.pos();
}

// Whole PointCloud2 (E3 layout) but the points.
inline size_t write_cloud(Writer &w, const char *frame_id, size_t fid_len, rplgpu_stamp_t stamp,
                          uint32_t n_points, rplgpu_cloud_layout_t *L) {
  static const struct { const char *name; uint32_t offset; } kFields[4] = {
      {"x", 0}, {"y", 4}, {"z", 8}, {"intensity", 12}};
  encapsulation(w);
  header(w, frame_id, fid_len, stamp);
  w.u32(1);  // height
  L->width_off = static_cast<uint32_t>(w.pos());
  w.u32(n_points);  // width
  w.u32(4);         // fields.size()
  for (const auto &f : kFields) {
    w.str(f.name, strlen(f.name));
    w.u32(f.offset);
    w.u8(7);   // sensor_msgs/PointField FLOAT32
    w.u32(1);  // count
  }
  w.u8(0);     // is_bigendian
  w.u32(16);   // point_step
  L->row_step_off = static_cast<uint32_t>(w.pos());
  w.u32(16u * n_points);
  L->data_len_off = static_cast<uint32_t>(w.pos());
  w.u32(16u * n_points);
  L->data_off = static_cast<uint32_t>(w.pos());
  w.skip(static_cast<size_t>(n_points) * 16);
  L->is_dense_off = static_cast<uint32_t>(w.pos());
  w.u8(1);  // is_dense
  L->total_len = static_cast<uint32_t>(w.pos());
  return w.pos();
}

}  // namespace rplmsg
