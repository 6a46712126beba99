/*
 * ref_node_gpu_shim.cpp — extern "C" door into the GENUINE RPlidarNode::publish_scan,
 * RPlidarNode::publish_cloud and RealLidarDriver::ascend_scan (the S1 seam of grab_scan_data) of
 * the reference PATCHED with integration/rplidar_node_gpu.patch (the GPU path behind the C ABI).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  No algorithm lives here.  Same construction as
 * ref_node_shim.cpp (a real RPlidarNode on the stand-in ROS headers of oracle/stubs/, the members
 * publish_scan reads set by hand), plus what on_configure / on_cleanup do in the patched node:
 * gpu_path_.configure(0, 8192) / gpu_path_.cleanup().  One node object lives across calls, as in
 * the real process (a handle is created once, not per scan).
 */
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <iostream>
#include <limits>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "rclcpp/rclcpp.hpp"
#include "rclcpp_lifecycle/lifecycle_node.hpp"
#include "rclcpp_lifecycle/lifecycle_publisher.hpp"
#include "sensor_msgs/msg/laser_scan.hpp"
#include "sensor_msgs/msg/point_cloud2.hpp"
#include "geometry_msgs/msg/transform_stamped.hpp"
#include "diagnostic_updater/diagnostic_updater.hpp"
#include "tf2/LinearMath/Quaternion.h"
#include "tf2_ros/static_transform_broadcaster.h"
#include "sl_lidar.h"
#include "sl_lidar_driver.h"

#define private public
#define protected public
#include "lidar_driver_wrapper.hpp"
#include "rplidar_node.hpp" /* the PATCHED header (oracle/_ref/patched/include comes first) */
#undef private
#undef protected

namespace {
std::unique_ptr<RPlidarNode> g_node;
std::string g_err;
}  // namespace

extern "C" {

struct refgpu_scan_meta {
  float angle_min, angle_max, angle_increment, time_increment;
  float scan_time, range_min, range_max;
  uint32_t count;
  int32_t published;
};

/* What the patched on_configure does with use_gpu = true.  Returns 1 when the GPU path is ready. */
int refgpu_open(int use_gpu) {
  g_node = std::make_unique<RPlidarNode>();
  g_node->params_.use_gpu = use_gpu != 0;
  if (g_node->params_.use_gpu && !g_node->gpu_path_.configure(0, 8192)) {
    g_err = g_node->gpu_path_.last_error();
    return 0;
  }
  return g_node->gpu_path_.ready() ? 1 : 0;
}
void refgpu_close(void) {
  if (g_node) g_node->gpu_path_.cleanup();
  g_node.reset();
}
/* ScanPath::set_min_samples: scans shorter than n are declined and take the node's CPU loop */
void refgpu_set_min_samples(unsigned n) {
  if (g_node) g_node->gpu_path_.set_min_samples(n);
}
int refgpu_ready(void) { return g_node && g_node->gpu_path_.ready() ? 1 : 0; }
const char *refgpu_last_error(void) {
  if (g_node) g_err = g_node->gpu_path_.last_error();
  return g_err.c_str();
}

/* driver_kind as in ref_node_shim.cpp: 0 Dummy, 1 Real OLD_TYPE, 2 Real NEW_TYPE. */
int refgpu_publish_scan(const void *nodes, size_t n, int driver_kind, int inverted,
                        int scan_processing, float cached_max_range, double scan_duration,
                        float *ranges, float *intensities, refgpu_scan_meta *meta) {
  std::memset(meta, 0, sizeof(*meta));
  if (!g_node) return -2;
  RPlidarNode &node = *g_node;
  node.params_.scan_processing = scan_processing != 0;
  node.params_.inverted = inverted != 0;
  node.params_.frame_id = "laser_frame";
  node.cached_current_max_range_ = cached_max_range;
  if (driver_kind == 0) {
    node.driver_ = std::make_unique<DummyLidarDriver>();
  } else {
    auto drv = std::make_unique<RealLidarDriver>();
    drv->profile_.protocol = (driver_kind == 2) ? ProtocolType::NEW_TYPE : ProtocolType::OLD_TYPE;
    node.driver_ = std::move(drv);
  }
  node.scan_pub_ =
      std::make_shared<rclcpp_lifecycle::LifecyclePublisher<sensor_msgs::msg::LaserScan>>();

  const auto *p = reinterpret_cast<const sl_lidar_response_measurement_node_hq_t *>(nodes);
  std::vector<sl_lidar_response_measurement_node_hq_t> vec(p, p + n);
  node.publish_scan(vec, rclcpp::Time(0), scan_duration);

  auto &pub = *node.scan_pub_;
  if (pub.publish_count == 0) return 0;
  const auto &msg = pub.last;
  meta->published = 1;
  meta->angle_min = msg.angle_min;
  meta->angle_max = msg.angle_max;
  meta->angle_increment = msg.angle_increment;
  meta->time_increment = msg.time_increment;
  meta->scan_time = msg.scan_time;
  meta->range_min = msg.range_min;
  meta->range_max = msg.range_max;
  meta->count = (uint32_t)msg.ranges.size();
  if (msg.ranges.size() != msg.intensities.size()) return -1;
  std::copy(msg.ranges.begin(), msg.ranges.end(), ranges);
  std::copy(msg.intensities.begin(), msg.intensities.end(), intensities);
  return 1;
}

/* S1: what the patched RealLidarDriver::grab_scan_data does with a grabbed scan
 * (src/lidar_driver_wrapper.cpp:328-329 in the unpatched file): ascend_scan, in place.  The driver
 * object is a genuine RealLidarDriver (its constructor makes the SDK driver; no connection is
 * needed for ascendScanData).  offer_gpu: the node's scan-loop line driver_->set_gpu_path(...). */
int refgpu_ascend_scan(void *nodes, size_t n, int offer_gpu) {
  if (!g_node) return -2;
  RealLidarDriver drv;
  LidarDriverInterface &iface = drv;
  iface.set_gpu_path(offer_gpu && g_node->gpu_path_.ready() ? &g_node->gpu_path_ : nullptr);
  drv.ascend_scan(reinterpret_cast<sl_lidar_response_measurement_node_hq_t *>(nodes), n);
  return 0;
}

/* ext: RPlidarNode::publish_cloud of the patched node (publish_cloud = true; leaf <= 0: no voxel
 * grid).  Returns the number of publications (0 or 1); xyzi receives width x 16 bytes. */
int refgpu_publish_cloud(const void *nodes, size_t n, int driver_kind, int inverted,
                         float cached_max_range, double leaf, float *xyzi, uint32_t *width,
                         uint32_t *layout_ok) {
  *width = 0;
  *layout_ok = 0;
  if (!g_node) return -2;
  RPlidarNode &node = *g_node;
  node.params_.inverted = inverted != 0;
  node.params_.frame_id = "laser_frame";
  node.params_.publish_cloud = true;
  node.params_.cloud_voxel_leaf = leaf;
  node.cached_current_max_range_ = cached_max_range;
  if (driver_kind == 0) {
    node.driver_ = std::make_unique<DummyLidarDriver>();
  } else {
    auto drv = std::make_unique<RealLidarDriver>();
    drv->profile_.protocol = (driver_kind == 2) ? ProtocolType::NEW_TYPE : ProtocolType::OLD_TYPE;
    node.driver_ = std::move(drv);
  }
  node.cloud_pub_ =
      std::make_shared<rclcpp_lifecycle::LifecyclePublisher<sensor_msgs::msg::PointCloud2>>();
  const auto *p = reinterpret_cast<const sl_lidar_response_measurement_node_hq_t *>(nodes);
  std::vector<sl_lidar_response_measurement_node_hq_t> vec(p, p + n);
  node.publish_cloud(vec, rclcpp::Time(0));
  auto &pub = *node.cloud_pub_;
  if (pub.publish_count == 0) return 0;
  const auto &m = pub.last;
  *width = m.width;
  static const char *const names[4] = {"x", "y", "z", "intensity"};
  bool ok = m.height == 1 && m.point_step == 16 && m.row_step == 16 * m.width && !m.is_bigendian &&
            m.is_dense && m.fields.size() == 4 && m.data.size() == (size_t)16 * m.width &&
            m.header.frame_id == "laser_frame";
  for (uint32_t f = 0; ok && f < 4; ++f)
    ok = m.fields[f].name == names[f] && m.fields[f].offset == 4 * f && m.fields[f].datatype == 7 &&
         m.fields[f].count == 1;
  *layout_ok = ok ? 1u : 0u;
  if (m.width) std::memcpy(xyzi, m.data.data(), (size_t)16 * m.width);
  return (int)pub.publish_count;
}

}  // extern "C"
