// rplgpu_host.hpp — C++ host side of the MI355X scan path, mirroring the reference's own
// call sites so that the node keeps its ROS 2 surface and only the per-sample loops move.
//
// Reference seams (citations relative to the reference tree):
//   S1  drv_->ascendScanData(buf, count)            src/lidar_driver_wrapper.cpp:329
//         -> rplgpu_host::ScanPath::ascendScanData(buf, count)
//   S3  RPlidarNode::publish_scan body :568-680      src/rplidar_node.cpp
//         -> rplgpu_host::ScanPath::fill_laser_scan(nodes, ..., scan_msg)   (everything up to,
//            but not including, scan_pub_->publish(scan_msg) at :682)
//   ext PointCloud2 (x, y, z, intensity FLOAT32; point_step 16) with clip / radius-outlier /
//         voxel grid -> rplgpu_host::ScanPath::fill_point_cloud2(nodes, ..., cloud_msg)
//   msg the same two, but ending in scan_pub_->publish(const rclcpp::SerializedMessage &):
//         the device results land by DMA inside the serialised (CDR) message, no typed message,
//         no std::vector fills, no middleware serialisation pass (include/rplgpu_msg.h)
//         -> rplgpu_host::ScanPath::fill_serialized_laser_scan / fill_serialized_point_cloud2
//   pre LIDARSampleDataUnpacker::onSampleData -> LIDARSampleDataListener callbacks
//         (src/sdk/src/dataunpacker/dataunpacker.h:48-88) and ScanDataHolder
//         (src/sdk/src/sl_lidar_driver.cpp:272-315), for RECORDED answer streams
//         -> rplgpu_host::ScanPath::replay_recording(ans, bytes, n, listener) /
//            rplgpu_host::ScanAssembler
//
// Header only, no ROS dependency: the message types are template parameters, so the same
// code compiles against sensor_msgs::msg::LaserScan / PointCloud2 in the node and against
// plain stand-ins in tests.  Everything goes through the C ABI of librplgpu.so
// (include/rplgpu.h); there is no CPU implementation here — on any error the functions return
// false and leave the message untouched, and the caller keeps its own CPU loop as fallback
// (INTEGRATION.md).
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "rplgpu.h"
#include "rplgpu_msg.h"

namespace rplgpu_host {

// Any 8-byte node type with the SDK layout (sl_lidar_response_measurement_node_hq_t,
// src/sdk/include/sl_lidar_cmd.h:272-278) can be passed; it is reinterpreted, never copied
// field by field.
template <class NodeT>
inline const rplgpu_node_t *as_nodes(const NodeT *p) {
  static_assert(sizeof(NodeT) == sizeof(rplgpu_node_t), "node must be the packed 8-byte SDK record");
  return reinterpret_cast<const rplgpu_node_t *>(p);
}
template <class NodeT>
inline rplgpu_node_t *as_nodes(NodeT *p) {
  static_assert(sizeof(NodeT) == sizeof(rplgpu_node_t), "node must be the packed 8-byte SDK record");
  return reinterpret_cast<rplgpu_node_t *>(p);
}

// The scalars publish_scan reads from the node (params_, driver_, cached range).
struct ScanConfig {
  bool is_new_protocol = false;     // RealLidarDriver && NEW_TYPE, src/rplidar_node.cpp:577-581
  bool inverted = false;            // params_.inverted, :646,:676
  bool scan_processing = true;      // params_.scan_processing, :632 (code default :280)
  float cached_current_max_range = 12.0f;  // :626
  // extensions (all off == the reference's behaviour)
  bool clip_enable = false;
  uint32_t q_min = 0;
  float range_min = 0.15f;
  bool ror_enable = false;
  float ror_radius = 0.10f;
  uint32_t ror_min_neighbors = 2;
  bool voxel_enable = false;
  float voxel_leaf = 0.05f;

  rplgpu_params_t to_params() const {
    rplgpu_params_t p;
    rplgpu_default_params(&p);
    p.is_new_protocol = is_new_protocol;
    p.inverted = inverted;
    p.scan_processing = scan_processing;
    p.clip_enable = clip_enable;
    p.q_min = q_min;
    p.range_min = range_min;
    p.range_max = cached_current_max_range;
    p.ror_enable = ror_enable;
    p.ror_radius = ror_radius;
    p.ror_min_neighbors = ror_min_neighbors;
    p.voxel_enable = voxel_enable;
    p.voxel_leaf = voxel_leaf;
    return p;
  }
};

// One per node instance (created in on_configure, destroyed in on_cleanup): owns one
// rplgpu handle, i.e. one HIP stream and its staging.  Not thread safe: the node calls it
// from its single scan thread (src/rplidar_node.cpp:222).
class ScanPath {
 public:
  ScanPath() = default;
  ScanPath(const ScanPath &) = delete;
  ScanPath &operator=(const ScanPath &) = delete;
  ~ScanPath() { cleanup(); }

  // on_configure (src/rplidar_node.cpp:116): bind to a device.  The SDK never hands out more
  // than 8192 nodes per scan (src/lidar_driver_wrapper.cpp:316-318).
  bool configure(int device_id = 0, uint32_t max_samples_per_scan = 8192) {
    cleanup();
    const int32_t rc = rplgpu_create(device_id, max_samples_per_scan, 1, &h_);
    if (rc != RPLGPU_OK) {
      last_error_ = std::string("rplgpu_create failed (") + std::to_string(rc) + "): " +
                    rplgpu_last_error(nullptr);
      h_ = nullptr;
      return false;
    }
    max_n_ = max_samples_per_scan;
    ranges_.resize(max_n_);
    intens_.resize(max_n_);
    xyzi_.resize(static_cast<size_t>(max_n_) * 4);
    return true;
  }
  // on_cleanup (:244)
  void cleanup() {
    if (h_) rplgpu_destroy(h_);
    h_ = nullptr;
  }
  bool ready() const { return h_ != nullptr; }
  // One scan per call pays a fixed ~21 us (launch + completion flag over PCIe) whatever its size;
  // the reference's loop needs ~7 us for the 360 samples of an A1 and overtakes the device path
  // below ~1700 samples.  Scans shorter than `n` are DECLINED:
  // fill_laser_scan / fill_point_cloud2 return false with last_error() set, i.e. the caller's own
  // CPU loop runs, exactly as after a device error.  0 (the default) declines nothing.
  void set_min_samples(uint32_t n) { min_samples_ = n; }
  uint32_t min_samples() const { return min_samples_; }
  const std::string &last_error() const { return last_error_; }

  // == sl::ILidarDriver::ascendScanData (src/sdk/include/sl_lidar_driver.h:477): in place,
  // returns the SDK's sl_result (0 = SL_RESULT_OK, 0x80008001 = SL_RESULT_OPERATION_FAIL when
  // every node is invalid, buffer untouched).  0x80008000 | other on a device error.
  template <class NodeT>
  uint32_t ascendScanData(NodeT *nodebuffer, size_t count) {
    uint32_t sl_result = 0x80008001u;
    last_error_.clear();
    if (count < min_samples_) {  // declined: the SDK's own loop is faster for a scan this small
      last_error_ = "scan below the configured minimum (CPU loop is faster)";
      return 0x80008002u;
    }
    if (!h_ || rplgpu_ascend(h_, as_nodes(nodebuffer), count, &sl_result) != RPLGPU_OK) {
      note_error();
      return 0x80008002u;  // SL_RESULT_OPERATION_TIMEOUT class: "did not happen"
    }
    return sl_result;
  }

  // == the body of RPlidarNode::publish_scan, src/rplidar_node.cpp:561-680.  Fills every
  // field of `scan_msg` that the reference fills except header.stamp / header.frame_id
  // (:620-621, the caller has them).  Returns false when the reference would have returned
  // without publishing (:561-563, :611-613) or on a device error (see last_error()).
  template <class NodeT, class LaserScanT>
  bool fill_laser_scan(const std::vector<NodeT> &nodes, const ScanConfig &cfg,
                       double scan_duration, LaserScanT &scan_msg) {
    if (nodes.empty()) return false;  // :561-563
    if (!h_ || nodes.size() > max_n_) return fail("scan larger than the configured capacity");
    if (nodes.size() < min_samples_) return fail("scan below the configured minimum (CPU loop is faster)");
    last_error_.clear();
    const rplgpu_params_t p = cfg.to_params();
    rplgpu_scan_meta_t meta;
    if (rplgpu_scan_to_laserscan(h_, as_nodes(nodes.data()), nodes.size(), &p, scan_duration,
                                 ranges_.data(), intens_.data(), &meta) != RPLGPU_OK)
      return note_error();
    if (!meta.published) return false;  // :611-613
    scan_msg.angle_min = meta.angle_min;              // :623
    scan_msg.angle_max = meta.angle_max;              // :624
    scan_msg.range_min = meta.range_min;              // :625
    scan_msg.range_max = meta.range_max;              // :626
    scan_msg.scan_time = meta.scan_time;              // :627
    scan_msg.angle_increment = meta.angle_increment;  // :635 / :666
    scan_msg.time_increment = meta.time_increment;    // :637 / :668
    scan_msg.ranges.assign(ranges_.begin(), ranges_.begin() + meta.count);
    scan_msg.intensities.assign(intens_.begin(), intens_.begin() + meta.count);
    return true;
  }

  // == publish_scan :561-682 for a publisher that sends serialised messages: `out` is an
  // rclcpp::SerializedMessage (or anything with reserve(size_t) and
  // get_rcl_serialized_message() -> {uint8_t *buffer; size_t buffer_length, buffer_capacity}).
  // On success `out` holds the complete CDR-serialised sensor_msgs/msg/LaserScan — header
  // stamp / frame_id (:620-621) included — ready for publish(out).  Same return rule as
  // fill_laser_scan.
  template <class NodeT, class SerializedT>
  bool fill_serialized_laser_scan(const std::vector<NodeT> &nodes, const ScanConfig &cfg,
                                  double scan_duration, const std::string &frame_id, int32_t sec,
                                  uint32_t nanosec, SerializedT &out) {
    if (nodes.empty()) return false;  // :561-563
    if (!h_ || nodes.size() > max_n_) return fail("scan larger than the configured capacity");
    if (nodes.size() < min_samples_) return fail("scan below the configured minimum (CPU loop is faster)");
    last_error_.clear();
    rplgpu_laserscan_layout_t L;
    if (rplgpu_msg_laserscan_layout(frame_id.size(), static_cast<uint32_t>(nodes.size()), &L))
      return fail("frame_id too long");
    out.reserve(L.total_len);  // worst case: every node becomes a beam
    auto &raw = out.get_rcl_serialized_message();
    const rplgpu_params_t p = cfg.to_params();
    rplgpu_scan_meta_t meta;
    size_t len = 0;
    if (rplgpu_scan_to_laserscan_msg(h_, as_nodes(nodes.data()), nodes.size(), &p, scan_duration,
                                     frame_id.c_str(), rplgpu_stamp_t{sec, nanosec}, raw.buffer,
                                     raw.buffer_capacity, &len, &meta) != RPLGPU_OK)
      return note_error();
    raw.buffer_length = len;
    return meta.published != 0;  // :611-613
  }

  // ext: the serialised sensor_msgs/msg/PointCloud2 of fill_point_cloud2.
  template <class NodeT, class SerializedT>
  bool fill_serialized_point_cloud2(const std::vector<NodeT> &nodes, const ScanConfig &cfg,
                                    const std::string &frame_id, int32_t sec, uint32_t nanosec,
                                    SerializedT &out) {
    if (nodes.empty()) return false;
    if (!h_ || nodes.size() > max_n_) return fail("scan larger than the configured capacity");
    if (nodes.size() < min_samples_) return fail("scan below the configured minimum (CPU loop is faster)");
    last_error_.clear();
    rplgpu_cloud_layout_t L;
    if (rplgpu_msg_cloud_layout(frame_id.size(), static_cast<uint32_t>(nodes.size()), &L))
      return fail("frame_id too long");
    out.reserve(L.total_len);
    auto &raw = out.get_rcl_serialized_message();
    const rplgpu_params_t p = cfg.to_params();
    size_t len = 0;
    uint32_t n_points = 0, status = 0;
    if (rplgpu_scan_to_cloud_msg(h_, as_nodes(nodes.data()), nodes.size(), &p, frame_id.c_str(),
                                 rplgpu_stamp_t{sec, nanosec}, raw.buffer, raw.buffer_capacity,
                                 &len, &n_points, &status) != RPLGPU_OK)
      return note_error();
    raw.buffer_length = len;
    return true;
  }

  // == feeding a whole recording of answer type `ans_type` to a fresh
  // sl::internal::LIDARSampleDataUnpacker (onSampleData, dataunpacker.h:79): `listener` gets the
  // calls the SDK's listener would get, in the same order — onHQNodeScanResetReq() and
  // onHQNodeDecoded(timestamp_uS, const node*) (dataunpacker.h:52-56; SlamtecLidarDriver's own
  // implementation is src/sdk/src/sl_lidar_driver.cpp:1645-1653).  Timestamps are 0: the SDK
  // stamps nodes with the decoding host's wall clock, which a recording does not contain.
  // `state` = {last sync bit, last dist_q2, 0, 0} carried between recordings of one sensor
  // (all zero for a fresh unpacker).  `sample_duration_us` = SlamtecLidarTimingDesc::
  // sample_duration_uS.  n_checksum_errors counts ERR_EVENT_ON_EXP_CHECKSUM_ERR events.
  template <class ListenerT>
  bool replay_recording(uint8_t ans_type, uint32_t sample_duration_us, const uint8_t *bytes,
                        size_t nbytes, ListenerT &listener, int32_t state[4],
                        uint32_t *n_checksum_errors = nullptr) {
    if (!h_) return fail("rplgpu handle not configured");
    const size_t S = rplgpu_frame_size(ans_type), npf = rplgpu_nodes_per_frame(ans_type);
    if (!S) return fail("unknown answer type");
    const size_t max_frames = nbytes / S;
    dec_nodes_.resize(max_frames * npf + 1);
    dec_resets_.resize(max_frames + 2);
    size_t n_nodes = 0, n_reset = 0;
    uint32_t n_err = 0;
    int32_t zero[4] = {0, 0, 0, 0};
    if (rplgpu_decode_stream(h_, ans_type, sample_duration_us, bytes, nbytes, state ? state : zero,
                             dec_nodes_.data(), dec_nodes_.size(), &n_nodes, dec_resets_.data(),
                             dec_resets_.size(), &n_reset, &n_err) != RPLGPU_OK)
      return note_error();
    if (n_checksum_errors) *n_checksum_errors = n_err;
    size_t r = 0;
    for (size_t i = 0; i <= n_nodes; ++i) {
      while (r < n_reset && dec_resets_[r] == i) {
        listener.onHQNodeScanResetReq();
        ++r;
      }
      if (i < n_nodes) listener.onHQNodeDecoded(0ull, &dec_nodes_[i]);
    }
    return true;
  }

  // ext: fills a sensor_msgs/PointCloud2-shaped message (fields x, y, z, intensity FLOAT32 at
  // offsets 0/4/8/12, point_step 16, height 1, is_dense, little endian).  `PointFieldT` is
  // sensor_msgs::msg::PointField (datatype 7 == FLOAT32).
  template <class NodeT, class PointCloud2T>
  bool fill_point_cloud2(const std::vector<NodeT> &nodes, const ScanConfig &cfg,
                         PointCloud2T &cloud_msg) {
    if (nodes.empty()) return false;
    if (!h_ || nodes.size() > max_n_) return fail("scan larger than the configured capacity");
    if (nodes.size() < min_samples_) return fail("scan below the configured minimum (CPU loop is faster)");
    last_error_.clear();
    const rplgpu_params_t p = cfg.to_params();
    uint32_t n_points = 0, status = 0;
    const int32_t rc = rplgpu_scan_to_cloud(h_, as_nodes(nodes.data()), nodes.size(), &p,
                                            xyzi_.data(), &n_points, &status);
    if (rc != RPLGPU_OK) return note_error();
    using FieldT = typename std::remove_reference<decltype(cloud_msg.fields)>::type::value_type;
    cloud_msg.fields.clear();
    static const char *const names[4] = {"x", "y", "z", "intensity"};
    for (uint32_t f = 0; f < 4; ++f) {
      FieldT pf;
      pf.name = names[f];
      pf.offset = 4 * f;
      pf.datatype = 7;  // sensor_msgs/PointField FLOAT32
      pf.count = 1;
      cloud_msg.fields.push_back(pf);
    }
    cloud_msg.height = 1;
    cloud_msg.width = n_points;
    cloud_msg.point_step = 16;
    cloud_msg.row_step = 16 * n_points;
    cloud_msg.is_bigendian = false;
    cloud_msg.is_dense = true;
    cloud_msg.data.resize(static_cast<size_t>(n_points) * 16);
    if (n_points) std::memcpy(cloud_msg.data.data(), xyzi_.data(), static_cast<size_t>(n_points) * 16);
    return true;
  }

 private:
  bool fail(const char *msg) {
    last_error_ = msg;
    return false;
  }
  bool note_error() {
    last_error_ = h_ ? rplgpu_last_error(h_) : "rplgpu handle not configured";
    return false;
  }
  rplgpu_handle_t h_ = nullptr;
  uint32_t max_n_ = 0;
  std::vector<float> ranges_, intens_, xyzi_;
  std::vector<rplgpu_node_t> dec_nodes_;
  std::vector<uint32_t> dec_resets_;
  std::string last_error_;
  uint32_t min_samples_ = 0;
};

// == ScanDataHolder<T> as SlamtecLidarDriver drives it (src/sdk/src/sl_lidar_driver.cpp:236-360,
// :1645-1653): a listener for replay_recording that assembles scans.  A node with flag bit 0
// closes the scan being built and opens the next one; nodes before the first sync node, or after
// a rewind until the next sync node, are discarded; a scan that reached max_count nodes keeps
// overwriting its last slot.  `on_scan(std::vector<rplgpu_node_t>&)` receives every completed
// scan (the SDK instead parks it for grabScanDataHq).  Host-side bookkeeping, no arithmetic: for
// batches the same rules run on the device (rplgpu_segment_batch_dev).
template <class OnScan>
class ScanAssembler {
 public:
  explicit ScanAssembler(OnScan on_scan, size_t max_count = 8192)
      : on_scan_(on_scan), max_count_(max_count ? max_count : 1) {}
  void onHQNodeScanResetReq() { cur_.clear(); }
  void onHQNodeDecoded(unsigned long long, const rplgpu_node_t *node) {
    if (node->flag & 1u) {
      if (!cur_.empty()) {
        on_scan_(cur_);
        cur_.clear();
      }
    } else if (cur_.empty()) {
      return;
    }
    if (cur_.size() >= max_count_) cur_.back() = *node;
    else cur_.push_back(*node);
  }

 private:
  OnScan on_scan_;
  size_t max_count_;
  std::vector<rplgpu_node_t> cur_;
};

// == DummyLidarDriver::grab_scan_data's synthetic ring (src/lidar_driver_wrapper.cpp:441-471),
// the reference's only fake backend (config 1): 360 nodes, one per degree, 2 m +- 0.5 m,
// quality 200, phase advancing 0.1 rad per scan.  `phase` is the caller's copy of the
// reference's function-static accumulator.
template <class NodeT>
inline void dummy_scan(float &phase, std::vector<NodeT> &nodes) {
  static_assert(sizeof(NodeT) == 8, "node must be the packed 8-byte SDK record");
  const int count = 360;
  nodes.clear();
  nodes.reserve(count);
  phase += 0.1f;  // :450
  for (int i = 0; i < count; ++i) {
    rplgpu_node_t nd;
    nd.angle_z_q14 = static_cast<uint16_t>(static_cast<float>(i) * 16384.0f / 90.0f);  // :456
    const float dist_meters = 2.0f + 0.5f * std::sin(static_cast<float>(i) * 3.141592f / 180.0f + phase);
    nd.dist_mm_q2 = static_cast<uint32_t>(dist_meters * 1000.0f * 4.0f);               // :463
    nd.quality = 200;                                                                  // :464
    nd.flag = 0;
    NodeT out;
    std::memcpy(&out, &nd, 8);
    nodes.push_back(out);
  }
}

}  // namespace rplgpu_host
