/*
 * ref_node_shim.cpp — extern "C" door into the GENUINE RPlidarNode::publish_scan.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  No algorithm lives here.  The shim
 * builds a real RPlidarNode (reference class, /root/reference/include/rplidar_node.hpp:97)
 * on top of the stand-in ROS headers in oracle/stubs/, sets the handful of members
 * publish_scan reads (params_.{scan_processing,inverted,frame_id},
 * cached_current_max_range_, driver_ — src/rplidar_node.cpp:577-581,621,626,632,646)
 * and calls the private method src/rplidar_node.cpp:558-683 itself.  The stub
 * publisher keeps the published LaserScan, which is copied out to plain arrays.
 *
 * `#define private public` is applied to the two reference headers only (after
 * every standard header is already included), so the shim may touch private
 * members; the reference translation units themselves are compiled untouched.
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
#include "geometry_msgs/msg/transform_stamped.hpp"
#include "diagnostic_updater/diagnostic_updater.hpp"
#include "tf2/LinearMath/Quaternion.h"
#include "tf2_ros/static_transform_broadcaster.h"
#include "sl_lidar.h"
#include "sl_lidar_driver.h"

#define private public
#define protected public
#include "lidar_driver_wrapper.hpp"
#include "rplidar_node.hpp"
#undef private
#undef protected

extern "C" {

/* Mirrors oracle.h's orc_scan_meta_t (kept separate: this TU must not depend on
 * the restatement it is used to validate). */
struct ref_scan_meta {
  float angle_min, angle_max, angle_increment, time_increment;
  float scan_time, range_min, range_max;
  uint32_t count;
  int32_t published;
};

/* driver_kind: 0 = DummyLidarDriver (is_new_protocol=false via failed dynamic_cast),
 *              1 = RealLidarDriver, protocol OLD_TYPE,
 *              2 = RealLidarDriver, protocol NEW_TYPE (is_new_protocol=true). */
int ref_publish_scan(const void *nodes, size_t n, int driver_kind, int inverted,
                     int scan_processing, float cached_max_range, double scan_duration,
                     float *ranges, float *intensities, ref_scan_meta *meta) {
  std::memset(meta, 0, sizeof(*meta));
  RPlidarNode node;
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

/* Genuine DummyLidarDriver::grab_scan_data (src/lidar_driver_wrapper.cpp:441-471).
 * NB: it advances a process-wide static phase and sleeps 100 ms per call. */
int ref_dummy_grab(void *nodes_out, size_t cap) {
  DummyLidarDriver drv;
  std::vector<sl_lidar_response_measurement_node_hq_t> v;
  if (!drv.grab_scan_data(v)) return -1;
  size_t m = std::min(cap, v.size());
  std::memcpy(nodes_out, v.data(), m * sizeof(v[0]));
  return (int)v.size();
}

/* WARMUP range rule, src/rplidar_node.cpp:391-396, evaluated by the same expression
 * on the genuine driver objects' hw limit (Dummy: 40 m, :439). */
float ref_dummy_hw_max_distance(void) {
  DummyLidarDriver drv;
  return drv.get_hw_max_distance();
}

}  // extern "C"
