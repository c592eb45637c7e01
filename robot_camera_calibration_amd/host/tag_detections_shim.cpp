// tag_detections_shim.cpp -- optional catkin target (host/CMakeLists.txt; NOT built in this repository's image: there
// is no ROS here).  Drop-in for the external detector node the reference launches with
//   roslaunch apriltag_ros continuous_detection.launch camera_name:=/cv_camera image_topic:=image_raw
// (real_preprocessing/README.md:65): subscribes <camera_name>/<image_topic>, calls the C ABI (include/rcc.h) and
// publishes "tag_detections" with exactly the fields the reference's consumer reads -- detections[i].size[0], .id[0],
// .pixel_corners_x/y[0..3] (real_preprocessing/src/corner_detections.cpp:46-54) -- and "tag_detections_image"
// (README.md:52,66), so corner_detection_node, camera_pose_node and opt_vis_node run unchanged.
//
// Two things follow from "unchanged":
//   * the published corners are pixels of the RAW image (undistort = 0): camera_pose_node hands them to
//     solvePnP together with kdistCoeffs (camera_pose.cpp:163) -- corners of an undistorted image would have the
//     distortion applied twice;
//   * the targets are square fiducials (RCC_TARGET_FIDUCIAL): camera_pose_node builds its object points as the
//     +-size/2 square (camera_pose.cpp:158-161), which a checkerboard's four outer corners are not.
//
// Parameters (private namespace): camera_name, image_topic (README.md:65), family_file (text file, one 36-bit
// payload per line in hex: robot_camera_calibration_amd/data/family36b.txt is the build's own), tag_size (metres),
// max_targets (result slots per frame), max_hamming.  Intrinsics come from the same rosparams camera_pose_node reads:
//   /camera_matrix/data (9 doubles, row-major), /distortion_coefficients/data (5 doubles)
// (real_preprocessing/src/camera_pose.cpp:59-64); where they are absent, from the K / D of <camera_name>/camera_info (upstream's
// detector pairs image + camera_info: README.md:64-65).
// Encodings: bgr8 (cv_camera's), rgb8, mono8; anything else is reported (throttled) and the frame skipped -- never read as BGR.
#include <ros/ros.h>
#include <sensor_msgs/CameraInfo.h>
#include <sensor_msgs/Image.h>
#include <apriltag_ros/AprilTagDetectionArray.h>
#include <fstream>
#include <string>
#include <vector>
#include "rcc.h"
#include "tag_detections_fill.h"

class RccDetectorNode {
 public:
  explicit RccDetectorNode(ros::NodeHandle& nh) : nh_(nh), h_(nullptr), w_(0), h_px_(0), max_targets_(64) {
    std::string cam, topic, family_file;
    ros::NodeHandle pnh("~");
    pnh.param<std::string>("camera_name", cam, "/cv_camera");
    pnh.param<std::string>("image_topic", topic, "image_raw");
    pnh.param<std::string>("family_file", family_file, "");
    pnh.param<double>("tag_size", tag_size_, 0.10);
    pnh.param<int>("max_targets", max_targets_, 64);
    pnh.param<int>("max_hamming", max_hamming_, 2);
    std::ifstream in(family_file.c_str());
    int bad_lines = 0;
    rcc_parse_family(in, family_, &bad_lines);        // never throws: malformed lines are skipped and counted
    if (bad_lines) ROS_ERROR("rcc_detector: %d malformed line(s) skipped in family file '%s'", bad_lines, family_file.c_str());
    if (family_.empty()) ROS_ERROR("rcc_detector: no tag family loaded from '%s'", family_file.c_str());
    det_.resize(max_targets_ > 0 ? max_targets_ : 1);
    pub_ = nh_.advertise<apriltag_ros::AprilTagDetectionArray>("tag_detections", 1);
    pub_img_ = nh_.advertise<sensor_msgs::Image>("tag_detections_image", 1);
    sub_ = nh_.subscribe(cam + "/" + topic, 1, &RccDetectorNode::onImage, this);   // queue 1, as the reference's consumer
    sub_info_ = nh_.subscribe(cam + "/camera_info", 1, &RccDetectorNode::onInfo, this);
  }
  ~RccDetectorNode() { rcc_destroy(h_); }

 private:
  void onInfo(const sensor_msgs::CameraInfo::ConstPtr& msg) {
    for (int i = 0; i < 9; ++i) info_K_[i] = msg->K[i];
    info_D_.assign(msg->D.begin(), msg->D.end());
    have_info_ = true;
  }

  bool ensureHandle(const sensor_msgs::Image& img, int pixfmt) {
    if (h_ && (int)img.width == w_ && (int)img.height == h_px_ && pixfmt == pixfmt_ && (int)img.step == step_) return true;
    rcc_destroy(h_);
    h_ = nullptr;
    if (family_.empty()) return false;
    rcc_config c;
    rcc_default_config(&c);
    c.width = img.width; c.height = img.height; c.stride_bytes = img.step;
    c.pixfmt = pixfmt;
    c.frame_bytes = (int64_t)img.step * img.height;
    std::vector<double> K, D;
    const bool params = nh_.getParam("/camera_matrix/data", K) && nh_.getParam("/distortion_coefficients/data", D);
    double Kd[9], Dd[5];
    const int src = rcc_pick_intrinsics(params ? &K : nullptr, params ? &D : nullptr, info_K_, info_D_.data(), (int)info_D_.size(), have_info_, Kd, Dd);
    if (!src) {
      ROS_ERROR_THROTTLE(5.0, "Camera intrinsics not loaded to parameter server!");   // the message of camera_pose.cpp:67 (and no camera_info seen either)
      return false;
    }
    if (src == 2) ROS_WARN_ONCE("rcc_detector: /camera_matrix/data not on the parameter server: intrinsics taken from camera_info");
    for (int i = 0; i < 9; ++i) c.K[i] = Kd[i];
    for (int i = 0; i < 5; ++i) c.D[i] = Dd[i];
    c.dist_model = RCC_DIST_PLUMB_BOB;
    c.undistort = 0;                                   // raw-image corners; D goes to solvePnP (camera_pose.cpp:163)
    c.target_kind = RCC_TARGET_FIDUCIAL;               // square targets of `size`: what camera_pose.cpp:158-161 assumes
    c.family_n = (int32_t)family_.size();
    c.family_codes = family_.data();
    c.tag_size = tag_size_;
    c.tag_max_hamming = max_hamming_;
    c.max_targets = (int32_t)det_.size();
    c.max_kept = RCC_MAX_KEPT_FIDUCIAL;
    c.max_candidates = 4096;
    c.xj_check = 0;
    c.batch_capacity = 1;
    int st = rcc_create(&c, &h_);
    if (st != RCC_OK) { ROS_ERROR("rcc_create: %s", rcc_status_string(st)); return false; }
    w_ = img.width; h_px_ = img.height; pixfmt_ = pixfmt; step_ = (int)img.step;
    return true;
  }

  void onImage(const sensor_msgs::Image::ConstPtr& msg) {
    const int pixfmt = rcc_pixfmt_of_encoding(msg->encoding);
    if (pixfmt < 0) {
      ROS_ERROR_THROTTLE(5.0, "rcc_detector: image encoding '%s' is not supported (bgr8, rgb8, mono8): frame skipped", msg->encoding.c_str());
      return;
    }
    const int ch = pixfmt == RCC_PIX_MONO8 ? 1 : 3;
    if (msg->step < msg->width * (unsigned)ch || msg->data.size() < (size_t)msg->step * msg->height) {
      ROS_ERROR_THROTTLE(5.0, "rcc_detector: malformed image (step %u, %zu bytes for %ux%u %s): frame skipped", msg->step, msg->data.size(), msg->width, msg->height, msg->encoding.c_str());
      return;
    }
    if (!ensureHandle(*msg, pixfmt)) return;
    int32_t n = 0;
    int st = rcc_detect_batch(h_, msg->data.data(), 1, RCC_MEM_HOST, det_.data(), &n, nullptr, nullptr);
    if (st != RCC_OK) { ROS_ERROR_THROTTLE(1.0, "rcc_detect_batch: %s", rcc_status_string(st)); return; }
    apriltag_ros::AprilTagDetectionArray out;
    rcc_fill_tag_detections<apriltag_ros::AprilTagDetectionArray, apriltag_ros::AprilTagDetection>(det_.data(), n, msg->header, out);
    pub_.publish(out);   // an empty array is skipped by the consumer (corner_detections.cpp:43)
    if (pub_img_.getNumSubscribers() > 0) {
      sensor_msgs::Image vis = *msg;
      rcc_draw_detections(vis.data.data(), vis.width, vis.height, vis.step, ch, det_.data(), n);
      pub_img_.publish(vis);
    }
  }

  ros::NodeHandle nh_;
  ros::Publisher pub_, pub_img_;
  ros::Subscriber sub_, sub_info_;
  rcc_handle* h_;
  int w_, h_px_, max_targets_, max_hamming_;
  int pixfmt_ = -1, step_ = 0;
  double info_K_[9] = { 0 };
  std::vector<double> info_D_;
  bool have_info_ = false;
  double tag_size_;
  std::vector<uint64_t> family_;
  std::vector<rcc_detection> det_;      // max_targets records: sized from the parameter, never a fixed stack array
};

int main(int argc, char** argv) {
  ros::init(argc, argv, "rcc_detector");
  ros::NodeHandle nh;
  RccDetectorNode node(nh);
  ros::spin();
  return 0;
}
