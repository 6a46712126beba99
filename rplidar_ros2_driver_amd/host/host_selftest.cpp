// host_selftest — drives the C++ host mirror (rplgpu_host.hpp) exactly the way the patched
// node would: grab_scan_data-style ascend, then the publish_scan body and the
// PointCloud2 extension, on a raw scan read from a file (8-byte nodes), and dumps the
// resulting messages as flat binary so that a test can compare them with the oracle.
//
//   host_selftest <nodes.bin> <out.bin> <is_new> <inverted> <scan_processing> <ascend 0|1> <cloud 0|1|2>
//   out.bin: u32 published, 7 x f32 meta, u32 count, ranges[count], intensities[count],
//            u32 sl_result, u32 n_points, xyzi[4*n_points]
//   host_selftest replay <ans_type> <sample_duration_us> <stream.bin> <out.bin>
//   out.bin: u32 n_nodes, u32 n_reset_calls, u32 n_err, u32 n_scans, nodes[n_nodes],
//            reset positions, then per scan: u32 len, nodes[len]   (replay_recording + ScanAssembler)
//   host_selftest serialized <nodes.bin> <out.bin> <is_new> <inverted> <scan_processing> <cloud 1|2>
//   out.bin: u32 published, u32 len_scan, LaserScan message bytes, u32 len_cloud, PointCloud2
//            message bytes   (frame_id "laser_frame", stamp 1727000000.123456789, duration 0.125)
//   no arguments: config-1 smoke run (3 Dummy scans through the whole path).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <type_traits>
#include <vector>

#include "rplgpu_host.hpp"

namespace {
struct PointField {
  std::string name;
  uint32_t offset = 0;
  uint8_t datatype = 0;
  uint32_t count = 0;
};
struct LaserScan {  // field names of sensor_msgs/msg/LaserScan
  float angle_min = 0, angle_max = 0, angle_increment = 0, time_increment = 0, scan_time = 0;
  float range_min = 0, range_max = 0;
  std::vector<float> ranges, intensities;
};
struct PointCloud2 {  // field names of sensor_msgs/msg/PointCloud2
  uint32_t height = 0, width = 0;
  std::vector<PointField> fields;
  bool is_bigendian = false;
  uint32_t point_step = 0, row_step = 0;
  std::vector<uint8_t> data;
  bool is_dense = false;
};
struct RclSerialized {  // the fields of rcl_serialized_message_t (rcutils_uint8_array_t) used here
  uint8_t *buffer = nullptr;
  size_t buffer_length = 0, buffer_capacity = 0;
};
class SerializedMessage {  // the two rclcpp::SerializedMessage members the host mirror calls
 public:
  void reserve(size_t capacity) {
    if (capacity > store_.size()) store_.resize(capacity);
    raw_.buffer = store_.data();
    raw_.buffer_capacity = store_.size();
  }
  RclSerialized &get_rcl_serialized_message() { return raw_; }
 private:
  std::vector<uint8_t> store_;
  RclSerialized raw_;
};
std::vector<rplgpu_node_t> read_nodes(const char *path, bool *ok) {
  std::vector<rplgpu_node_t> nodes;
  *ok = false;
  std::FILE *f = std::fopen(path, "rb");
  if (!f) return nodes;
  std::fseek(f, 0, SEEK_END);
  const long bytes = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  nodes.resize(static_cast<size_t>(bytes) / 8);
  *ok = nodes.empty() || std::fread(nodes.data(), 8, nodes.size(), f) == nodes.size();
  std::fclose(f);
  return nodes;
}
}  // namespace

int main(int argc, char **argv) {
  rplgpu_host::ScanPath path;
  if (!path.configure(0, 32768)) {
    std::fprintf(stderr, "configure failed: %s\n", path.last_error().c_str());
    return 2;
  }
  if (argc == 6 && std::string(argv[1]) == "replay") {
    struct Recorder {  // what SlamtecLidarDriver's listener would see, plus the scan assembly
      std::vector<rplgpu_node_t> nodes;
      std::vector<uint32_t> resets;
      std::vector<std::vector<rplgpu_node_t>> scans;
    } rec;
    auto on_scan = [&](std::vector<rplgpu_node_t> &s) { rec.scans.push_back(s); };
    rplgpu_host::ScanAssembler<decltype(on_scan)> assembler(on_scan, 8192);
    struct Tee {
      Recorder &r;
      rplgpu_host::ScanAssembler<decltype(on_scan)> &a;
      void onHQNodeScanResetReq() {
        r.resets.push_back(static_cast<uint32_t>(r.nodes.size()));
        a.onHQNodeScanResetReq();
      }
      void onHQNodeDecoded(unsigned long long ts, const rplgpu_node_t *n) {
        r.nodes.push_back(*n);
        a.onHQNodeDecoded(ts, n);
      }
    } tee{rec, assembler};
    std::FILE *f = std::fopen(argv[4], "rb");
    if (!f) return 4;
    std::fseek(f, 0, SEEK_END);
    const long nb = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> bytes(static_cast<size_t>(nb));
    if (nb && std::fread(bytes.data(), 1, bytes.size(), f) != bytes.size()) return 4;
    std::fclose(f);
    int32_t state[4] = {0, 0, 0, 0};
    uint32_t n_err = 0;
    if (!path.replay_recording(static_cast<uint8_t>(std::strtoul(argv[2], nullptr, 0)),
                               static_cast<uint32_t>(std::atoi(argv[3])), bytes.data(), bytes.size(),
                               tee, state, &n_err)) {
      std::fprintf(stderr, "replay failed: %s\n", path.last_error().c_str());
      return 5;
    }
    std::FILE *o = std::fopen(argv[5], "wb");
    if (!o) return 6;
    const uint32_t hdr[4] = {static_cast<uint32_t>(rec.nodes.size()), static_cast<uint32_t>(rec.resets.size()),
                             n_err, static_cast<uint32_t>(rec.scans.size())};
    std::fwrite(hdr, 4, 4, o);
    std::fwrite(rec.nodes.data(), 8, rec.nodes.size(), o);
    std::fwrite(rec.resets.data(), 4, rec.resets.size(), o);
    for (auto &s : rec.scans) {
      const uint32_t len = static_cast<uint32_t>(s.size());
      std::fwrite(&len, 4, 1, o);
      std::fwrite(s.data(), 8, s.size(), o);
    }
    std::fclose(o);
    return 0;
  }
  if (argc == 8 && std::string(argv[1]) == "serialized") {
    bool ok = false;
    const std::vector<rplgpu_node_t> nodes = read_nodes(argv[2], &ok);
    if (!ok) return 4;
    rplgpu_host::ScanConfig cfg;
    cfg.is_new_protocol = std::atoi(argv[4]) != 0;
    cfg.inverted = std::atoi(argv[5]) != 0;
    cfg.scan_processing = std::atoi(argv[6]) != 0;
    cfg.cached_current_max_range = 40.0f;
    SerializedMessage scan_msg, cloud_msg;
    scan_msg.reserve(16);
    cloud_msg.reserve(16);
    const bool published = path.fill_serialized_laser_scan(nodes, cfg, 0.125, "laser_frame",
                                                           1727000000, 123456789u, scan_msg);
    cfg.clip_enable = true;
    cfg.voxel_enable = std::atoi(argv[7]) == 2;
    if (!path.fill_serialized_point_cloud2(nodes, cfg, "laser_frame", 1727000000, 123456789u,
                                           cloud_msg) && !nodes.empty()) {
      std::fprintf(stderr, "cloud failed: %s\n", path.last_error().c_str());
      return 5;
    }
    std::FILE *o = std::fopen(argv[3], "wb");
    if (!o) return 6;
    const uint32_t pub = published ? 1u : 0u;
    const auto &a = scan_msg.get_rcl_serialized_message();
    const auto &c = cloud_msg.get_rcl_serialized_message();
    const uint32_t la = static_cast<uint32_t>(a.buffer_length), lc = static_cast<uint32_t>(c.buffer_length);
    std::fwrite(&pub, 4, 1, o);
    std::fwrite(&la, 4, 1, o);
    std::fwrite(a.buffer, 1, la, o);
    std::fwrite(&lc, 4, 1, o);
    std::fwrite(c.buffer, 1, lc, o);
    std::fclose(o);
    return 0;
  }
  if (argc < 8) {
    float phase = 0.0f;
    std::vector<rplgpu_node_t> nodes;
    rplgpu_host::ScanConfig cfg;
    cfg.cached_current_max_range = 40.0f;  // Dummy hw limit, src/lidar_driver_wrapper.cpp:439
    for (int s = 0; s < 3; ++s) {
      rplgpu_host::dummy_scan(phase, nodes);
      LaserScan msg;
      if (!path.fill_laser_scan(nodes, cfg, 0.1, msg) || msg.ranges.size() != 360) {
        std::fprintf(stderr, "dummy scan %d failed: %s\n", s, path.last_error().c_str());
        return 3;
      }
      std::printf("dummy scan %d: %zu beams, ranges[0]=%.6f intensities[0]=%.1f\n", s,
                  msg.ranges.size(), msg.ranges[0], msg.intensities[0]);
    }
    return 0;
  }
  std::FILE *f = std::fopen(argv[1], "rb");
  if (!f) return 4;
  std::fseek(f, 0, SEEK_END);
  const long bytes = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  std::vector<rplgpu_node_t> nodes(static_cast<size_t>(bytes) / 8);
  if (!nodes.empty() && std::fread(nodes.data(), 8, nodes.size(), f) != nodes.size()) return 4;
  std::fclose(f);

  rplgpu_host::ScanConfig cfg;
  cfg.is_new_protocol = std::atoi(argv[3]) != 0;
  cfg.inverted = std::atoi(argv[4]) != 0;
  cfg.scan_processing = std::atoi(argv[5]) != 0;
  cfg.cached_current_max_range = 40.0f;
  uint32_t sl_result = 0;
  if (std::atoi(argv[6])) sl_result = path.ascendScanData(nodes.data(), nodes.size());  // S1

  LaserScan msg;
  const bool published = path.fill_laser_scan(nodes, cfg, 0.125, msg);  // S3
  PointCloud2 cloud;
  const int cloud_mode = std::atoi(argv[7]);
  if (cloud_mode) {
    cfg.clip_enable = true;
    cfg.voxel_enable = cloud_mode == 2;
    if (!path.fill_point_cloud2(nodes, cfg, cloud) && !nodes.empty()) {
      std::fprintf(stderr, "cloud failed: %s\n", path.last_error().c_str());
      return 5;
    }
  }
  std::FILE *o = std::fopen(argv[2], "wb");
  if (!o) return 6;
  const uint32_t pub = published ? 1u : 0u, count = static_cast<uint32_t>(msg.ranges.size());
  const float meta[7] = {msg.angle_min, msg.angle_max, msg.angle_increment, msg.time_increment,
                         msg.scan_time, msg.range_min, msg.range_max};
  std::fwrite(&pub, 4, 1, o);
  std::fwrite(meta, 4, 7, o);
  std::fwrite(&count, 4, 1, o);
  std::fwrite(msg.ranges.data(), 4, count, o);
  std::fwrite(msg.intensities.data(), 4, count, o);
  std::fwrite(&sl_result, 4, 1, o);
  std::fwrite(&cloud.width, 4, 1, o);
  std::fwrite(cloud.data.data(), 1, cloud.data.size(), o);
  // the (possibly ascended) nodes, so the caller can check S1 as well
  std::fwrite(nodes.data(), 8, nodes.size(), o);
  std::fclose(o);
  return 0;
}
