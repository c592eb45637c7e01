// mock_spin.cpp -- TEST INFRASTRUCTURE (see ros/ros.h): ros::spin() of the stand-in.  Plays a bag file to the node's subscribers and
// writes what the node publishes as text, one line per message:
//   A <seq> <ndet>                                  a "tag_detections" array ...
//   D <id> <size> x0 y0 x1 y1 x2 y2 x3 y3 tx ty tz  ... and its detections (bl, br, tr, tl), %.17g
//   V <seq> <pixels that differ from the input>     a "tag_detections_image"
//   S <seq>                                         an image message the node published nothing for (skipped / no handle)
// Bag file (written by tests/test_shim_node_run.py): "RCCBAG1\n", then records
//   'I' K[9] (f64) nD (u32) D[nD] (f64)                                      -> <camera>/camera_info
//   'F' seq width height step (u32) nenc (u32) enc[nenc] nbytes (u64) data   -> <camera>/<image topic>
#include <ros/ros.h>
#include <sensor_msgs/CameraInfo.h>
#include <sensor_msgs/Image.h>
#include <apriltag_ros/AprilTagDetectionArray.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

namespace {
bool rd(FILE* f, void* p, size_t n) { return std::fread(p, 1, n, f) == n; }
}

void ros::spin()
{
  auto& bus = ros::mock::Bus::get();
  const std::string bag = bus.special.count("bag") ? bus.special["bag"] : "", outp = bus.special.count("out") ? bus.special["out"] : "";
  const std::string cam = bus.private_params.count("camera_name") ? bus.private_params["camera_name"] : "/cv_camera";
  const std::string topic = bus.private_params.count("image_topic") ? bus.private_params["image_topic"] : "image_raw";
  FILE* in = std::fopen(bag.c_str(), "rb");
  FILE* out = std::fopen(outp.c_str(), "w");
  if (!in || !out) { std::fprintf(stderr, "mock spin: cannot open bag '%s' / out '%s'\n", bag.c_str(), outp.c_str()); std::exit(3); }
  char magic[8];
  if (!rd(in, magic, 8) || std::memcmp(magic, "RCCBAG1\n", 8) != 0) { std::fprintf(stderr, "mock spin: not a bag\n"); std::exit(3); }
  const sensor_msgs::Image* current = nullptr;
  int published = 0;
  bus.sinks["tag_detections"] = [&](const void* p) {
    const auto& a = *static_cast<const apriltag_ros::AprilTagDetectionArray*>(p);
    std::fprintf(out, "A %u %zu\n", a.header.seq, a.detections.size());
    for (const auto& d : a.detections) {
      std::fprintf(out, "D %d %.17g", d.id.at(0), d.size.at(0));
      for (int k = 0; k < 4; ++k) std::fprintf(out, " %.17g %.17g", d.pixel_corners_x.at(k), d.pixel_corners_y.at(k));
      std::fprintf(out, " %.17g %.17g %.17g\n", d.pose.pose.pose.position.x, d.pose.pose.pose.position.y, d.pose.pose.pose.position.z);
    }
    ++published;
  };
  bus.sinks["tag_detections_image"] = [&](const void* p) {
    const auto& v = *static_cast<const sensor_msgs::Image*>(p);
    size_t diff = 0;
    if (current && v.data.size() == current->data.size()) {
      const int ch = v.encoding == "mono8" ? 1 : 3;
      for (uint32_t y = 0; y < v.height; ++y)
        for (uint32_t x = 0; x < v.width; ++x)
          diff += std::memcmp(&v.data[(size_t)y * v.step + (size_t)x * ch], &current->data[(size_t)y * v.step + (size_t)x * ch], ch) != 0;
    }
    std::fprintf(out, "V %u %zu\n", v.header.seq, diff);
  };
  for (;;) {
    char kind;
    if (!rd(in, &kind, 1)) break;
    if (kind == 'I') {
      auto m = std::make_shared<sensor_msgs::CameraInfo>();
      uint32_t nd = 0;
      if (!rd(in, m->K.data(), 72) || !rd(in, &nd, 4) || nd > 64) break;
      m->D.resize(nd);
      if (nd && !rd(in, m->D.data(), 8 * (size_t)nd)) break;
      auto it = bus.subscribers.find(cam + "/camera_info");
      sensor_msgs::CameraInfo::ConstPtr cp = m;
      if (it != bus.subscribers.end()) it->second(&cp);
    } else if (kind == 'F') {
      auto m = std::make_shared<sensor_msgs::Image>();
      uint32_t hdr[4], nenc = 0;
      uint64_t nbytes = 0;
      if (!rd(in, hdr, 16) || !rd(in, &nenc, 4) || nenc > 64) break;
      m->header.seq = hdr[0]; m->width = hdr[1]; m->height = hdr[2]; m->step = hdr[3];
      m->encoding.resize(nenc);
      if (nenc && !rd(in, &m->encoding[0], nenc)) break;
      if (!rd(in, &nbytes, 8) || nbytes > (1ull << 31)) break;
      m->data.resize((size_t)nbytes);
      if (nbytes && !rd(in, m->data.data(), (size_t)nbytes)) break;
      sensor_msgs::Image::ConstPtr cp = m;
      current = m.get();
      const int before = published;
      auto it = bus.subscribers.find(cam + "/" + topic);
      if (it != bus.subscribers.end()) it->second(&cp);
      if (published == before) std::fprintf(out, "S %u\n", m->header.seq);
      current = nullptr;
    } else {
      break;
    }
  }
  bus.sinks.clear();        // they refer to this function's locals
  std::fclose(in);
  std::fclose(out);
}
