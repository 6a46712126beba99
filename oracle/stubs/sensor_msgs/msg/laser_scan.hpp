// Stand-in for <sensor_msgs/msg/laser_scan.hpp> — see oracle/stubs/README.md.
// Field names and scalar types follow the public sensor_msgs/LaserScan.msg definition
// (float32 scalars, float32[] arrays).
#pragma once
#include <string>
#include <vector>

#include "rclcpp/rclcpp.hpp"

namespace std_msgs_stub {
struct Header {
  rclcpp::Time stamp;
  std::string frame_id;
};
}  // namespace std_msgs_stub

namespace sensor_msgs { namespace msg {
struct LaserScan {
  std_msgs_stub::Header header;
  float angle_min = 0, angle_max = 0, angle_increment = 0;
  float time_increment = 0, scan_time = 0;
  float range_min = 0, range_max = 0;
  std::vector<float> ranges;
  std::vector<float> intensities;
};
}}  // namespace sensor_msgs::msg
