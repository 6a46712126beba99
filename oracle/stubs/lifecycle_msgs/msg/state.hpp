// Stand-in for <lifecycle_msgs/msg/state.hpp> — see oracle/stubs/README.md.
#pragma once
#include <cstdint>
namespace lifecycle_msgs { namespace msg {
struct State {
  static constexpr uint8_t PRIMARY_STATE_UNCONFIGURED = 1;
  static constexpr uint8_t PRIMARY_STATE_INACTIVE = 2;
  static constexpr uint8_t PRIMARY_STATE_ACTIVE = 3;
};
}}  // namespace lifecycle_msgs::msg
