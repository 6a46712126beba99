// Stand-in for <rclcpp/rclcpp.hpp> — see oracle/stubs/README.md. Test infrastructure only.
#pragma once
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <memory>
#include <string>
#include <vector>

namespace rcl_interfaces { namespace msg {
struct SetParametersResult {
  bool successful = true;
  std::string reason;
};
}}  // namespace rcl_interfaces::msg

namespace rclcpp {

struct NodeOptions {};

class Duration {
 public:
  explicit Duration(int64_t ns = 0) : ns_(ns) {}
  double seconds() const { return static_cast<double>(ns_) * 1e-9; }
 private:
  int64_t ns_;
};

class Time {
 public:
  Time() : ns_(0) {}
  explicit Time(int64_t ns) : ns_(ns) {}
  int64_t nanoseconds() const { return ns_; }
  Duration operator-(const Time &o) const { return Duration(ns_ - o.ns_); }
 private:
  int64_t ns_;
};

struct Clock {
  using SharedPtr = std::shared_ptr<Clock>;
};

struct Logger {};

enum class ParameterType { PARAMETER_NOT_SET, PARAMETER_BOOL, PARAMETER_INTEGER, PARAMETER_DOUBLE, PARAMETER_STRING };

class Parameter {
 public:
  Parameter() = default;
  const std::string &get_name() const { return name_; }
  ParameterType get_type() const { return type_; }
  int64_t as_int() const { return i_; }
  bool as_bool() const { return b_; }
  std::string as_string() const { return s_; }
 private:
  std::string name_;
  ParameterType type_ = ParameterType::PARAMETER_NOT_SET;
  int64_t i_ = 0;
  bool b_ = false;
  std::string s_;
};

class QoS {
 public:
  explicit QoS(size_t depth) : depth_(depth) {}
  QoS &reliable() { return *this; }
  QoS &best_effort() { return *this; }
  QoS &durability_volatile() { return *this; }
 private:
  size_t depth_;
};

inline bool ok() { return false; }  // the scan thread is never meant to run in the oracle

}  // namespace rclcpp

// Logging macros swallow their arguments (still evaluated for side effects: none).
#define RCLCPP_INFO(logger, ...) do { (void)(logger); } while (0)
#define RCLCPP_WARN(logger, ...) do { (void)(logger); } while (0)
#define RCLCPP_ERROR(logger, ...) do { (void)(logger); } while (0)
#define RCLCPP_WARN_THROTTLE(logger, clock, period, ...) do { (void)(logger); } while (0)
