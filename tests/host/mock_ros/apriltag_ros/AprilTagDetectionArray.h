// stand-in (see ../ros/ros.h): the fields of the fork's message the reference's consumer reads
// (real_preprocessing/src/corner_detections.cpp:43-54) + the upstream pose field the node fills
#pragma once
#include <memory>
#include <vector>
#include <sensor_msgs/Image.h>
namespace geometry_msgs {
struct Point { double x = 0, y = 0, z = 0; };
struct Pose { Point position; };
struct PoseWithCovariance { Pose pose; };
struct PoseWithCovarianceStamped { std_msgs::Header header; PoseWithCovariance pose; };
}  // namespace geometry_msgs
namespace apriltag_ros {
struct AprilTagDetection {
  std::vector<int> id;
  std::vector<double> size;
  std::vector<double> pixel_corners_x, pixel_corners_y;
  geometry_msgs::PoseWithCovarianceStamped pose;
};
struct AprilTagDetectionArray {
  typedef std::shared_ptr<const AprilTagDetectionArray> ConstPtr;
  std_msgs::Header header;
  std::vector<AprilTagDetection> detections;
};
}  // namespace apriltag_ros
