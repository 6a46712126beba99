// Stand-in for <sensor_msgs/msg/point_cloud2.hpp> — see oracle/stubs/README.md.
// Field names and types follow the public sensor_msgs/PointCloud2.msg and PointField.msg
// definitions (uint32 height/width/point_step/row_step, uint8[] data, bool flags).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "sensor_msgs/msg/laser_scan.hpp"  // std_msgs_stub::Header

namespace sensor_msgs { namespace msg {
struct PointField {
  std::string name;
  uint32_t offset = 0;
  uint8_t datatype = 0;  // 7 == FLOAT32
  uint32_t count = 0;
};
struct PointCloud2 {
  std_msgs_stub::Header header;
  uint32_t height = 0, width = 0;
  std::vector<PointField> fields;
  bool is_bigendian = false;
  uint32_t point_step = 0, row_step = 0;
  std::vector<uint8_t> data;
  bool is_dense = false;
};
}}  // namespace sensor_msgs::msg
