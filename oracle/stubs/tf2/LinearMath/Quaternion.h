// Stand-in for <tf2/LinearMath/Quaternion.h> — see oracle/stubs/README.md.
#pragma once
namespace tf2 {
class Quaternion {
 public:
  void setRPY(double, double, double) {}
  double x() const { return 0; }
  double y() const { return 0; }
  double z() const { return 0; }
  double w() const { return 1; }
};
}  // namespace tf2
