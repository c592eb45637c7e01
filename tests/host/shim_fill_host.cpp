// Host check of the message filling of the optional ROS node (robot_camera_calibration_amd/host/
// tag_detections_fill.h) with stand-in message structs that carry only the field names the reference's
// consumer reads (real_preprocessing/src/corner_detections.cpp:43-54).  No ROS involved.
#include <vector>
#include <cstring>
#include <sstream>
#include "../../include/rcc.h"
#include "../../robot_camera_calibration_amd/host/tag_detections_fill.h"

namespace mock {
struct Header { unsigned seq; double stamp; };
struct Point { double x, y, z; };
struct Pose { Point position; };
struct PoseWithCov { Pose pose; };
struct PoseStamped { Header header; PoseWithCov pose; };
struct Detection {
  std::vector<int> id;
  std::vector<double> size;
  std::vector<double> pixel_corners_x, pixel_corners_y;
  PoseStamped pose;
};
struct Array { Header header; std::vector<Detection> detections; };
}  // namespace mock

// out: per detection 1 + 1 + 8 + 3 doubles = id, size, x0..x3, y0..y3, position; returns number of detections,
// and replays the consumer's own reads: int(pixel_corners_x[n]) (corner_detections.cpp:53)
extern "C" int shimfill_roundtrip(const rcc_detection* det, int n, unsigned seq, double* out, int* as_int)
{
  mock::Array msg;
  mock::Header h{ seq, 1.5 };
  rcc_fill_tag_detections<mock::Array, mock::Detection>(det, n, h, msg);
  if (msg.header.seq != seq) return -1;
  for (size_t i = 0; i < msg.detections.size(); ++i) {
    const mock::Detection& d = msg.detections[i];
    if (d.id.size() != 1 || d.size.size() != 1 || d.pixel_corners_x.size() != 4 || d.pixel_corners_y.size() != 4) return -2;
    double* o = out + 13 * i;
    o[0] = d.id[0]; o[1] = d.size[0];
    for (int k = 0; k < 4; ++k) { o[2 + k] = d.pixel_corners_x[k]; o[6 + k] = d.pixel_corners_y[k]; as_int[8 * i + k] = int(d.pixel_corners_x[k]); as_int[8 * i + 4 + k] = int(d.pixel_corners_y[k]); }
    o[10] = d.pose.pose.pose.position.x; o[11] = d.pose.pose.pose.position.y; o[12] = d.pose.pose.pose.position.z;
  }
  return (int)msg.detections.size();
}

// the overlay of "tag_detections_image" (real_preprocessing/README.md:52,66)
extern "C" void shimfill_draw(unsigned char* img, int width, int height, int step, int channels, const rcc_detection* det, int n)
{
  rcc_draw_detections(img, width, height, step, channels, det, n);
}

// the node's family_file parser: returns the number of codes (at most cap are copied out), *bad = malformed lines skipped
extern "C" int shimfill_parse_family(const char* text, unsigned long long* out, int cap, int* bad)
{
  std::istringstream in(text);
  std::vector<uint64_t> codes;
  rcc_parse_family(in, codes, bad);
  for (size_t i = 0; i < codes.size() && (int)i < cap; ++i) out[i] = codes[i];
  return (int)codes.size();
}

// the node's encoding policy and intrinsics fallback (tag_detections_fill.h)
extern "C" int shimfill_pixfmt(const char* encoding) { return rcc_pixfmt_of_encoding(encoding ? std::string(encoding) : std::string()); }
extern "C" int shimfill_pick_intrinsics(const double* pK, int npK, const double* pD, int npD, const double* infoK, const double* infoD, int ninfoD,
                                        int have_info, double* K9, double* D5)
{
  std::vector<double> vK, vD;
  if (pK) vK.assign(pK, pK + npK);
  if (pD) vD.assign(pD, pD + npD);
  return rcc_pick_intrinsics(pK ? &vK : nullptr, pD ? &vD : nullptr, infoK, infoD, ninfoD, have_info != 0, K9, D5);
}
