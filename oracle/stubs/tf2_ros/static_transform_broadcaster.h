// Stand-in for <tf2_ros/static_transform_broadcaster.h> — see oracle/stubs/README.md.
#pragma once
#include "geometry_msgs/msg/transform_stamped.hpp"
namespace tf2_ros {
class StaticTransformBroadcaster {
 public:
  template <class NodeT>
  explicit StaticTransformBroadcaster(NodeT *) {}
  void sendTransform(const geometry_msgs::msg::TransformStamped &) {}
};
}  // namespace tf2_ros
