// Stand-in for <rclcpp_lifecycle/lifecycle_publisher.hpp> — see oracle/stubs/README.md.
#pragma once
#include <memory>

namespace rclcpp_lifecycle {

// Captures the most recent message so the oracle shim can read it back.
template <class MsgT>
class LifecyclePublisher {
 public:
  using SharedPtr = std::shared_ptr<LifecyclePublisher<MsgT>>;
  void publish(const MsgT &msg) {
    last = msg;
    ++publish_count;
  }
  MsgT last{};
  unsigned long publish_count = 0;
};

}  // namespace rclcpp_lifecycle
