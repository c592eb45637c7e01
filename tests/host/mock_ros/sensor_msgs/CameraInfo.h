// stand-in (see ../ros/ros.h): the fields of sensor_msgs/CameraInfo the node reads
#pragma once
#include <array>
#include <memory>
#include <vector>
#include "Image.h"
namespace sensor_msgs {
struct CameraInfo {
  typedef std::shared_ptr<const CameraInfo> ConstPtr;
  std_msgs::Header header;
  std::vector<double> D;
  std::array<double, 9> K;
};
}  // namespace sensor_msgs
