// Stand-in for the handful of roscpp declarations host/tag_detections_shim.cpp uses -- TEST INFRASTRUCTURE: there is no ROS in
// this image.  Two uses:
//   * tests/test_tagmap_yaml.py::test_ros_shim_node_compiles_against_stand_in_headers runs `g++ -fsyntax-only` over the node with
//     these on the include path: the node's own code is well-formed C++ against the shapes of the API it calls;
//   * tests/test_shim_node_run.py builds the node + mock_spin.cpp into an executable and RUNS it (round 4): a process-local "bus"
//     delivers the messages of a small bag file to the node's subscribers and hands what the node publishes to the test -- the
//     compiled host above the C ABI, on the GPU, without roscpp.  Parameters come from the command line in ROS's own remapping
//     syntax (_name:=value for the private namespace, /name:=v1,v2,... for the global vectors camera_pose_node reads).
// It proves nothing about roscpp itself (transport, queues, timing).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <type_traits>
#include <vector>
namespace ros {
namespace mock {
struct Bus {
  std::map<std::string, std::string> private_params;                    // "_name:=value"
  std::map<std::string, std::vector<double>> global_vectors;            // "/name:=v1,v2,..."
  std::map<std::string, std::string> special;                           // "__name:=value" (the harness's own: bag, out)
  std::map<std::string, std::function<void(const void*)>> subscribers;  // topic -> callback; the argument points at an M::ConstPtr
  std::map<std::string, std::function<void(const void*)>> sinks;        // topic -> the harness's reader; the argument points at an M
  static Bus& get() { static Bus b; return b; }
};
inline void parse(const std::string& s, std::string& v) { v = s; }
inline void parse(const std::string& s, double& v) { v = std::strtod(s.c_str(), nullptr); }
inline void parse(const std::string& s, int& v) { v = (int)std::strtol(s.c_str(), nullptr, 10); }
}  // namespace mock
struct Publisher {
  std::string topic;
  template <class M> void publish(const M& m) const {
    auto& b = mock::Bus::get();
    auto it = b.sinks.find(topic);
    if (it != b.sinks.end()) it->second(&m);
  }
  unsigned getNumSubscribers() const { return (unsigned)mock::Bus::get().sinks.count(topic); }
};
struct Subscriber {};
struct NodeHandle {
  bool priv = false;
  NodeHandle() {}
  explicit NodeHandle(const std::string& ns) : priv(ns == "~") {}
  template <class T> void param(const std::string& name, T& v, const T& d) const {
    v = d;
    if (!priv) return;
    auto& p = mock::Bus::get().private_params;
    auto it = p.find(name);
    if (it != p.end()) mock::parse(it->second, v);
  }
  bool getParam(const std::string& name, std::vector<double>& v) const {
    auto& g = mock::Bus::get().global_vectors;
    auto it = g.find(name);
    if (it == g.end()) return false;
    v = it->second;
    return true;
  }
  template <class M> Publisher advertise(const std::string& topic, unsigned) { Publisher p; p.topic = topic; return p; }
  // roscpp deduces the message type from the callback's argument; the node relies on that form
  template <class A, class T> Subscriber subscribe(const std::string& topic, unsigned, void (T::*fn)(A), T* obj) {
    typedef typename std::remove_cv<typename std::remove_reference<A>::type>::type Ptr;
    mock::Bus::get().subscribers[topic] = [fn, obj](const void* p) { (obj->*fn)(*static_cast<const Ptr*>(p)); };
    return Subscriber();
  }
};
inline void init(int& argc, char** argv, const std::string&) {
  auto& b = mock::Bus::get();
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    const size_t k = a.find(":=");
    if (k == std::string::npos) continue;
    const std::string name = a.substr(0, k), val = a.substr(k + 2);
    if (name.size() > 2 && name[0] == '_' && name[1] == '_') b.special[name.substr(2)] = val;
    else if (name.size() > 1 && name[0] == '_') b.private_params[name.substr(1)] = val;
    else if (!name.empty() && name[0] == '/') {
      std::vector<double> v;
      std::stringstream ss(val);
      for (std::string tok; std::getline(ss, tok, ',');) v.push_back(std::strtod(tok.c_str(), nullptr));
      b.global_vectors[name] = v;
    }
  }
}
void spin();      // tests/host/mock_ros/mock_spin.cpp: plays the bag named by __bag:=, writes what is published to __out:=
}  // namespace ros
#define ROS_ERROR(...) (std::fprintf(stderr, "[ERROR] "), std::fprintf(stderr, __VA_ARGS__), std::fprintf(stderr, "\n"))
#define ROS_ERROR_THROTTLE(period, ...) (std::fprintf(stderr, "[ERROR] "), std::fprintf(stderr, __VA_ARGS__), std::fprintf(stderr, "\n"))
#define ROS_WARN_ONCE(...) (std::fprintf(stderr, "[WARN] "), std::fprintf(stderr, __VA_ARGS__), std::fprintf(stderr, "\n"))
