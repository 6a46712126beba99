// Stand-in for <geometry_msgs/msg/transform_stamped.hpp> — see oracle/stubs/README.md.
#pragma once
#include <string>

#include "sensor_msgs/msg/laser_scan.hpp"

namespace geometry_msgs { namespace msg {
struct Vector3 { double x = 0, y = 0, z = 0; };
struct QuaternionMsg { double x = 0, y = 0, z = 0, w = 1; };
struct Transform { Vector3 translation; QuaternionMsg rotation; };
struct TransformStamped {
  std_msgs_stub::Header header;
  std::string child_frame_id;
  Transform transform;
};
}}  // namespace geometry_msgs::msg
