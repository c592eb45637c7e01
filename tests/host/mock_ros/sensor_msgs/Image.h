// stand-in (see ../ros/ros.h): the fields of sensor_msgs/Image the node reads
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>
namespace std_msgs { struct Header { uint32_t seq = 0; double stamp = 0; std::string frame_id; }; }
namespace sensor_msgs {
struct Image {
  typedef std::shared_ptr<const Image> ConstPtr;
  std_msgs::Header header;
  uint32_t height = 0, width = 0, step = 0;
  std::string encoding;
  uint8_t is_bigendian = 0;
  std::vector<uint8_t> data;
};
}  // namespace sensor_msgs
