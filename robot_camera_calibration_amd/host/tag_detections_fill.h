// tag_detections_fill.h -- rcc_detection records -> the messages the reference's consumer subscribes to.
//
// Kept apart from the ROS node (tag_detections_shim.cpp) so that it can be compiled and tested without
// ROS: the message types are template parameters; the only thing assumed of them is the field names the
// reference reads -- detections (vector), and per element size[0], id[0], pixel_corners_x[0..3],
// pixel_corners_y[0..3] (real_preprocessing/src/corner_detections.cpp:43-54) -- plus the upstream pose
// field (never read by the reference).  Corner order bl, br, tr, tl (camera_pose.cpp:123-126).
#pragma once
#include <cerrno>
#include <cstdint>
#include <cstdlib>
#include <istream>
#include <string>
#include <vector>
#include "rcc.h"

// The node's `family_file`: one code word per line, hexadecimal (with or without 0x), 36-bit row-major payload, MSB
// first (rcc_config.family_codes; data/family_from_apriltag.py writes this format from an apriltag family description).
// Blank lines and lines starting with '#' are skipped.  Nothing here throws: a malformed line (not hexadecimal, trailing
// garbage, more than 36 bits) is counted in *bad_lines and skipped, so a typo in the file cannot terminate the node at start-up.
inline void rcc_parse_family(std::istream& in, std::vector<uint64_t>& codes, int* bad_lines)
{
  int bad = 0;
  for (std::string line; std::getline(in, line);) {
    size_t b = line.find_first_not_of(" \t\r");
    if (b == std::string::npos || line[b] == '#') continue;
    const char* p = line.c_str() + b;
    char* end = nullptr;
    errno = 0;
    const unsigned long long v = std::strtoull(p, &end, 16);
    bool ok = end != p && errno == 0 && (v >> 36) == 0 && *p != '-' && *p != '+';
    for (const char* q = end; ok && *q; ++q) ok = (*q == ' ' || *q == '\t' || *q == '\r');
    if (ok) codes.push_back((uint64_t)v); else ++bad;
  }
  if (bad_lines) *bad_lines = bad;
}

// sensor_msgs/Image::encoding -> rcc_config.pixfmt, or -1 for an encoding the detector does not take.  cv_camera publishes "bgr8"
// (real_preprocessing/README.md:25); "rgb8" and "mono8" are converted by the ingest pass itself (same luma, byte order of the
// encoding).  Everything else -- bgra8 / rgba8 (4 bytes per pixel), yuv422, 16-bit and Bayer encodings -- is REFUSED: read as
// BGR8 it would be mis-converted silently.  The node reports it (throttled) and skips the frame.
inline int rcc_pixfmt_of_encoding(const std::string& enc)
{
  if (enc == "bgr8" || enc == "8UC3") return RCC_PIX_BGR8;
  if (enc == "rgb8") return RCC_PIX_RGB8;
  if (enc == "mono8" || enc == "8UC1") return RCC_PIX_MONO8;
  return -1;
}

// Intrinsics of the detector: the rosparams camera_pose_node reads (/camera_matrix/data, 9 doubles row-major, and
// /distortion_coefficients/data, >= 5 doubles: real_preprocessing/src/camera_pose.cpp:59-64) when both are well-formed, else the
// K (9) and D (plumb_bob, >= 5; fewer are padded with zeros) of the last sensor_msgs/CameraInfo seen on <camera_name>/camera_info
// (upstream pairs image + camera_info: real_preprocessing/README.md:64-65).  Returns 1 = rosparams, 2 = camera_info, 0 = neither
// (K[0], K[4] must be positive either way).  Never throws.
inline int rcc_pick_intrinsics(const std::vector<double>* pK, const std::vector<double>* pD,
                               const double* infoK, const double* infoD, int ninfoD, bool have_info,
                               double K[9], double D[5])
{
  if (pK && pD && pK->size() == 9 && pD->size() >= 5 && (*pK)[0] > 0.0 && (*pK)[4] > 0.0) {
    for (int i = 0; i < 9; ++i) K[i] = (*pK)[i];
    for (int i = 0; i < 5; ++i) D[i] = (*pD)[i];
    return 1;
  }
  if (have_info && infoK && infoK[0] > 0.0 && infoK[4] > 0.0) {
    for (int i = 0; i < 9; ++i) K[i] = infoK[i];
    for (int i = 0; i < 5; ++i) D[i] = (infoD && i < ninfoD) ? infoD[i] : 0.0;
    return 2;
  }
  return 0;
}

template <class ArrayMsg, class DetMsg, class Header>
inline void rcc_fill_tag_detections(const rcc_detection* det, int n, const Header& header, ArrayMsg& out)
{
  out.header = header;
  out.detections.clear();
  for (int i = 0; i < n; ++i) {
    DetMsg d;
    d.id.push_back(det[i].id);                 // read as id[0]   (corner_detections.cpp:49)
    d.size.push_back(det[i].size);             // read as size[0] (corner_detections.cpp:48)
    for (int k = 0; k < 4; ++k) {              // read as pixel_corners_x/y[n], cast to int (corner_detections.cpp:53-54)
      d.pixel_corners_x.push_back(det[i].corners[k][0]);
      d.pixel_corners_y.push_back(det[i].corners[k][1]);
    }
    d.pose.header = header;                    // upstream field; the reference never reads it
    d.pose.pose.pose.position.x = det[i].tvec[0];
    d.pose.pose.pose.position.y = det[i].tvec[1];
    d.pose.pose.pose.position.z = det[i].tvec[2];
    out.detections.push_back(d);
  }
}

// "tag_detections_image" (real_preprocessing/README.md:52,66: what the user watches in image_view to see that the tags
// are found): the outline of every detection drawn into a copy of the input frame, bl->br->tr->tl->bl, the first edge
// (bl->br) brighter so that the orientation shows.  Plain integer line rasteriser, clipped to the image; `channels`
// 1 (mono8) or 3 (bgr8).
inline void rcc_draw_detections(uint8_t* img, int width, int height, int step, int channels, const rcc_detection* det, int n)
{
  for (int i = 0; i < n; ++i) {
    for (int e = 0; e < 4; ++e) {
      int x0 = (int)det[i].corners[e][0], y0 = (int)det[i].corners[e][1];
      const int x1 = (int)det[i].corners[(e + 1) & 3][0], y1 = (int)det[i].corners[(e + 1) & 3][1];
      const int dx = std::abs(x1 - x0), sx = x0 < x1 ? 1 : -1, dy = -std::abs(y1 - y0), sy = y0 < y1 ? 1 : -1;
      int err = dx + dy;
      for (int guard = 0; guard < 4 * (width + height); ++guard) {
        if (x0 >= 0 && x0 < width && y0 >= 0 && y0 < height) {
          uint8_t* p = img + (size_t)y0 * step + (size_t)x0 * channels;
          if (channels == 3) { p[0] = 0; p[1] = e == 0 ? 255 : 160; p[2] = e == 0 ? 0 : 255; }   // BGR: green first edge, orange others
          else p[0] = e == 0 ? 255 : 0;
        }
        if (x0 == x1 && y0 == y1) break;
        const int e2 = 2 * err;
        if (e2 >= dy) { err += dy; x0 += sx; }
        if (e2 <= dx) { err += dx; y0 += sy; }
      }
    }
  }
}
