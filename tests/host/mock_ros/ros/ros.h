// Stand-in for the handful of roscpp declarations host/tag_detections_shim.cpp uses -- TEST INFRASTRUCTURE: there is no ROS in
// this image, so the node has never met a compiler.  tests/test_tagmap_yaml.py::test_ros_shim_node_compiles_against_stand_in_headers
// runs `g++ -fsyntax-only` over the node with these on the include path: it proves the node's own code is well-formed C++
// against the shapes of the API it calls (names, argument kinds), nothing about roscpp itself.
#pragma once
#include <cstdio>
#include <map>
#include <memory>
#include <string>
#include <vector>
namespace ros {
struct Publisher {
  template <class M> void publish(const M&) const {}
  unsigned getNumSubscribers() const { return 0; }
};
struct Subscriber {};
struct NodeHandle {
  NodeHandle() {}
  explicit NodeHandle(const std::string&) {}
  template <class T> void param(const std::string&, T& v, const T& d) const { v = d; }
  template <class T> bool getParam(const std::string&, T&) const { return false; }
  template <class M> Publisher advertise(const std::string&, unsigned) { return Publisher(); }
  template <class M, class T> Subscriber subscribe(const std::string&, unsigned, void (T::*)(const typename M::ConstPtr&), T*) { return Subscriber(); }
  // roscpp deduces M from the callback's argument; the node relies on that form
  template <class A, class T> Subscriber subscribe(const std::string&, unsigned, void (T::*)(A), T*) { return Subscriber(); }
};
inline void init(int&, char**, const std::string&) {}
inline void spin() {}
}  // namespace ros
#define ROS_ERROR(...) std::fprintf(stderr, __VA_ARGS__)
#define ROS_ERROR_THROTTLE(period, ...) std::fprintf(stderr, __VA_ARGS__)
#define ROS_WARN_ONCE(...) std::fprintf(stderr, __VA_ARGS__)
