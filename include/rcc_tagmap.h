/*
 * rcc_tagmap.h -- host-side C ABI for the two "next" rows of SURVEY.md 8(f):
 *   N1  the tag-map builder of real_preprocessing/src/camera_pose.cpp (worldLoad :71-80,
 *       tagCalc :176-203, fileReader :207-246, unknownFilepoll :249-263, fileStream :267-285),
 *       fed from memory with the poses rcc_detect_batch / rcc_solve_pnp_batch return instead of
 *       re-parsing detections_N.yaml for every tag (camera_pose.cpp:134);
 *   N2  the YAML wire formats: detections_N.yaml (corner_detections.cpp:18-39), the appended
 *       world_T_camera block (camera_pose.cpp:83-100) and targets.yaml (camera_pose.cpp:103-129),
 *       byte for byte (std::to_string's 6 decimals, the reference's spacing).
 * Pure host code (no GPU); implemented in robot_camera_calibration_amd/host/tagmap.cpp.
 * No exception crosses this boundary and no pointer is read unchecked: rcc_tagmap_add_frame returns -1 (map unchanged) on a NULL
 * argument, n < 1 or memory exhaustion; the writers return 0 and write an empty string on a NULL input array or exhaustion.
 */
#ifndef RCC_TAGMAP_H_
#define RCC_TAGMAP_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* frame status, the reference's own codes (camera_pose.cpp:10-14) */
enum { RCC_MAP_WORLD_PRES = 0, RCC_MAP_KNOWN_TAG = 1, RCC_MAP_UNKNOWN = 2 };

typedef struct rcc_tagmap rcc_tagmap;
rcc_tagmap* rcc_tagmap_create(void);
void rcc_tagmap_destroy(rcc_tagmap* m);

/* One captured frame = one detections_N.yaml of the reference: n tags with ids, sizes and the
 * solver's cam_T_tag pose (rvec, tvec as solvePnP returns them, camera_pose.cpp:163).  Frames are
 * numbered in call order; frame 0 defines the world tag (its first tag, camera_pose.cpp:74).
 * Returns the status; *has_pose = 1 and world_T_cam (row-major 4x4) when the frame was localised
 * now.  A frame with only unknown tags is kept and retried after later frames, newest first, as
 * unknownFilepoll does (camera_pose.cpp:249-263). */
int rcc_tagmap_add_frame(rcc_tagmap* m, int32_t n, const int32_t* ids, const double* sizes,
                         const double* rvec, const double* tvec, double* world_T_cam, int32_t* has_pose);
/* world_T_camera of frame f if it has been localised (possibly by a later retry); returns 1/0 */
int rcc_tagmap_frame_pose(const rcc_tagmap* m, int32_t frame, double* world_T_cam);
int32_t rcc_tagmap_ntags(const rcc_tagmap* m);
int rcc_tagmap_tag(const rcc_tagmap* m, int32_t index, int32_t* id, double* size, double* world_T_tag);
int32_t rcc_tagmap_pending(const rcc_tagmap* m);   /* frames still waiting for a known tag */

/* N2 writers: write into buf (NUL-terminated) and return the length needed (excluding NUL);
 * if the return value >= cap the text was truncated. */
size_t rcc_yaml_detections(char* buf, size_t cap, int32_t n, const int32_t* ids, const double* sizes,
                           const int32_t* corners /* n x 4 x 2, bl br tr tl, already cast to int */);
size_t rcc_yaml_world_T_camera(char* buf, size_t cap, const double* world_T_cam);
size_t rcc_yaml_targets(const rcc_tagmap* m, char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
