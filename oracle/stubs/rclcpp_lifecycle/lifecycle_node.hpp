// Stand-in for <rclcpp_lifecycle/lifecycle_node.hpp> — see oracle/stubs/README.md.
#pragma once
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "rclcpp/rclcpp.hpp"
#include "rclcpp_lifecycle/lifecycle_publisher.hpp"

namespace rclcpp_lifecycle {

class State {
 public:
  explicit State(uint8_t id = 0) : id_(id) {}
  uint8_t id() const { return id_; }
 private:
  uint8_t id_;
};

namespace node_interfaces {
struct LifecycleNodeInterface {
  enum class CallbackReturn : uint8_t { SUCCESS = 97, FAILURE = 98, ERROR = 99 };
};
}  // namespace node_interfaces

struct OnSetParametersCallbackHandleStub {
  using SharedPtr = std::shared_ptr<OnSetParametersCallbackHandleStub>;
};

class LifecycleNode {
 public:
  using CallbackReturn = node_interfaces::LifecycleNodeInterface::CallbackReturn;
  using OnSetParametersCallbackHandle = OnSetParametersCallbackHandleStub;

  LifecycleNode(const std::string &, const rclcpp::NodeOptions &) {}
  virtual ~LifecycleNode() = default;

  virtual CallbackReturn on_configure(const State &) { return CallbackReturn::SUCCESS; }
  virtual CallbackReturn on_activate(const State &) { return CallbackReturn::SUCCESS; }
  virtual CallbackReturn on_deactivate(const State &) { return CallbackReturn::SUCCESS; }
  virtual CallbackReturn on_cleanup(const State &) { return CallbackReturn::SUCCESS; }
  virtual CallbackReturn on_shutdown(const State &) { return CallbackReturn::SUCCESS; }

  template <class T, class D>
  void declare_parameter(const std::string &, const D &) {}
  template <class T>
  bool get_parameter(const std::string &, T &) const { return false; }
  template <class T>
  bool get_parameter_or(const std::string &, T &value, const T &alt) const {
    value = alt;
    return false;
  }
  rclcpp::Logger get_logger() const { return rclcpp::Logger{}; }
  rclcpp::Time now() const { return rclcpp::Time(0); }
  rclcpp::Clock::SharedPtr get_clock() { return std::make_shared<rclcpp::Clock>(); }
  State get_current_state() const { return State(0); }

  template <class F>
  OnSetParametersCallbackHandle::SharedPtr add_on_set_parameters_callback(F &&) {
    return std::make_shared<OnSetParametersCallbackHandle>();
  }
  template <class MsgT>
  typename LifecyclePublisher<MsgT>::SharedPtr create_publisher(const std::string &,
                                                                const rclcpp::QoS &) {
    return std::make_shared<LifecyclePublisher<MsgT>>();
  }
};

}  // namespace rclcpp_lifecycle
