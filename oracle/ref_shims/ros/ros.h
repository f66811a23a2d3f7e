// ros::Time only (key type of VoxgraphSubmap's pose history).  TEST INFRASTRUCTURE.
#ifndef ORACLE_REF_SHIMS_ROS_ROS_H_
#define ORACLE_REF_SHIMS_ROS_ROS_H_
#include <cstdint>
namespace ros {
class Time {
 public:
  Time() : nsec_(0) {}
  explicit Time(double t) : nsec_(static_cast<uint64_t>(t * 1e9)) {}
  uint64_t toNSec() const { return nsec_; }
  bool operator<(const Time& o) const { return nsec_ < o.nsec_; }

 private:
  uint64_t nsec_;
};
}  // namespace ros
#endif
