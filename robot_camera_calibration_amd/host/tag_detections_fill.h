// tag_detections_fill.h -- rcc_detection records -> the message the reference's consumer subscribes to.
//
// Kept apart from the ROS node (tag_detections_shim.cpp) so that it can be compiled and tested without
// ROS: the message types are template parameters; the only thing assumed of them is the field names the
// reference reads -- detections (vector), and per element size[0], id[0], pixel_corners_x[0..3],
// pixel_corners_y[0..3] (real_preprocessing/src/corner_detections.cpp:43-54) -- plus the upstream pose
// field (never read by the reference).  Corner order bl, br, tr, tl (camera_pose.cpp:123-126).
#pragma once
#include "rcc.h"

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
