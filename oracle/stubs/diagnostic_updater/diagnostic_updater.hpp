// Stand-in for <diagnostic_updater/diagnostic_updater.hpp> — see oracle/stubs/README.md.
#pragma once
#include <cstdint>
#include <string>

namespace diagnostic_msgs { namespace msg {
struct DiagnosticStatus {
  static constexpr uint8_t OK = 0;
  static constexpr uint8_t WARN = 1;
  static constexpr uint8_t ERROR = 2;
  static constexpr uint8_t STALE = 3;
};
}}  // namespace diagnostic_msgs::msg

namespace diagnostic_updater {
class DiagnosticStatusWrapper {
 public:
  void summary(uint8_t, const std::string &) {}
  template <class T>
  void add(const std::string &, const T &) {}
};
class Updater {
 public:
  template <class NodeT>
  explicit Updater(NodeT *) {}
  void setHardwareID(const std::string &) {}
  template <class T>
  void add(const std::string &, T *, void (T::*)(DiagnosticStatusWrapper &)) {}
};
}  // namespace diagnostic_updater
