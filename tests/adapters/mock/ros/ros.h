// TEST INFRASTRUCTURE ONLY -- see ocs2_core/Types.h in this directory.
#pragma once
#include <map>
#include <string>
#include <vector>
#include <cstdio>
#define ROS_WARN_STREAM(msg) do { ros::warnings().push_back([&]() { std::ostringstream os_; os_ << msg; return os_.str(); }()); } while (0)
#include <sstream>
namespace ros {
inline std::vector<std::string>& warnings() { static std::vector<std::string> w; return w; }   // test hook: what rosconsole would have printed
class Publisher {};
class NodeHandle {
 public:
  NodeHandle() = default;
  NodeHandle(const NodeHandle& parent, const std::string& ns) : ns_(parent.ns_ + "/" + ns) {}
  const std::string& getNamespace() const { return ns_; }
  std::string ns_;
  template <class M> Publisher advertise(const std::string& topic, unsigned queue) { advertised.push_back(topic); (void)queue; return Publisher(); }
  std::vector<std::string> advertised;
};
namespace param {
inline std::map<std::string, std::string>& store() { static std::map<std::string, std::string> s; return s; }
inline bool get(const std::string& key, std::string& out) { auto it = store().find(key); if (it == store().end()) return false; out = it->second; return true; }
}  // namespace param
}  // namespace ros
namespace ocs2_msgs { struct mpc_observation {}; }
namespace qm_msgs { struct ee_state {}; }
