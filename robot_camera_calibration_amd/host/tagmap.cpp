// tagmap.cpp -- N1 tag-map builder and N2 YAML writers (include/rcc_tagmap.h).  Host only.
//
// Mirrors real_preprocessing/src/camera_pose.cpp's PoseSystem on in-memory frames:
//   world tag = first tag of frame 0 (:74), its transform the identity (:75-77);
//   a frame is referenced through the world tag if it is visible (:231-235), otherwise through the
//   LAST known tag in its list (the loop keeps overwriting known_tag_in_file, :236-240);
//   w_T_cam = w_T_tag(known) * tag_T_cam(known) (:184), new tags get w_T_cam * inverse(tag_T_cam) (:198);
//   frames with only unknown tags are deferred (:278-281) and retried newest first, and a retry is
//   taken only on KNOWN_TAG, not on WORLD_PRES (:256).
// tag_T_cam = inverse(cam_T_tag) with cam_T_tag = [Rodrigues(rvec) | tvec] (:164-172).
// Rotation matrix -> vector for the YAML output follows cv::Rodrigues via
// the product's own pnp_core.h.
//
// Boundary rules (include/rcc.h:17-18 hold here too): nothing thrown inside -- std::bad_alloc from a vector or a string --
// crosses the C ABI: every entry point catches everything and reports failure through its return value (add_frame: -1 with the
// map unchanged -- all storage the call can need is reserved before the first element is touched; the writers: 0 and an empty
// string), and every pointer argument is checked before it is read.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/rcc_tagmap.h"
#include "../csrc/pnp_core.h"

namespace {
struct M4 { double v[16]; };
M4 ident() { M4 m; for (int i = 0; i < 16; ++i) m.v[i] = (i % 5 == 0) ? 1.0 : 0.0; return m; }
M4 mul(const M4& a, const M4& b) {
  M4 c;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += a.v[i * 4 + k] * b.v[k * 4 + j]; c.v[i * 4 + j] = s; }
  return c;
}
M4 rigid_inverse(const M4& a) {   // [R t; 0 1]^-1 = [R^T -R^T t; 0 1]
  M4 c = ident();
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) c.v[i * 4 + j] = a.v[j * 4 + i];
  for (int i = 0; i < 3; ++i) c.v[i * 4 + 3] = -(c.v[i * 4] * a.v[3] + c.v[i * 4 + 1] * a.v[7] + c.v[i * 4 + 2] * a.v[11]);
  return c;
}
M4 from_rt(const double* r, const double* t) {
  double R[9];
  rccpnp::rodrigues_v2m(r, R, nullptr);
  M4 m = ident();
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) m.v[i * 4 + j] = R[i * 3 + j]; m.v[i * 4 + 3] = t[i]; }
  return m;
}
struct Frame { std::vector<int32_t> ids; std::vector<double> sizes; std::vector<M4> tag_T_cam; bool has_pose = false; M4 w_T_cam; };
}  // namespace

struct rcc_tagmap {
  std::vector<Frame> frames;
  int32_t world_tag = -1;
  std::vector<int32_t> ids;
  std::vector<double> sizes;
  std::vector<M4> w_T_tag;
  std::vector<int32_t> unreferenced;

  int find(int32_t id) const { auto it = std::find(ids.begin(), ids.end(), id); return it == ids.end() ? -1 : (int)(it - ids.begin()); }

  // fileReader (:207-246)
  int reader(int fn, int& known) {
    Frame& f = frames[fn];
    int status = RCC_MAP_UNKNOWN;
    if (fn == 0) {   // worldLoad (:71-80)
      known = 0;
      world_tag = f.ids[0];
      w_T_tag.push_back(ident()); ids.push_back(world_tag); sizes.push_back(f.sizes[0]);
      return RCC_MAP_WORLD_PRES;
    }
    for (int i = 0; i < (int)f.ids.size(); ++i) {
      if (f.ids[i] == world_tag) { known = i; return RCC_MAP_WORLD_PRES; }
      if (find(f.ids[i]) >= 0) { known = i; status = RCC_MAP_KNOWN_TAG; }
    }
    return status;
  }
  // tagCalc (:176-203)
  void calc(int fn, int known) {
    Frame& f = frames[fn];
    const int idx = find(f.ids[known]);
    f.w_T_cam = mul(w_T_tag[idx], f.tag_T_cam[known]);
    f.has_pose = true;
    for (int i = 0; i < (int)f.ids.size(); ++i) {
      if (i != known && find(f.ids[i]) < 0) {
        w_T_tag.push_back(mul(f.w_T_cam, rigid_inverse(f.tag_T_cam[i])));
        ids.push_back(f.ids[i]);
        sizes.push_back(f.sizes[i]);
      }
    }
  }
  // unknownFilepoll (:249-263)
  void poll() {
    for (int k = (int)unreferenced.size() - 1; k >= 0; --k) {
      int known = 0;
      if (reader(unreferenced[k], known) == RCC_MAP_KNOWN_TAG) {
        calc(unreferenced[k], known);
        unreferenced.erase(unreferenced.begin() + k);
      }
    }
  }
};

static std::string to6(double v) { return std::to_string(v); }       // "%f": 6 decimals, as the reference

static void rot_to_rvec(const M4& m, double r[3]) {
  const double R[9] = { m.v[0], m.v[1], m.v[2], m.v[4], m.v[5], m.v[6], m.v[8], m.v[9], m.v[10] };
  rccpnp::rodrigues_m2v(R, r);
}

static size_t emit(const std::string& s, char* buf, size_t cap) {
  if (buf && cap) { size_t n = std::min(s.size(), cap - 1); memcpy(buf, s.data(), n); buf[n] = 0; }
  return s.size();
}

static size_t emit_nothing(char* buf, size_t cap) { if (buf && cap) buf[0] = 0; return 0; }

extern "C" {

rcc_tagmap* rcc_tagmap_create(void) { return new (std::nothrow) rcc_tagmap(); }
void rcc_tagmap_destroy(rcc_tagmap* m) { delete m; }

int rcc_tagmap_add_frame(rcc_tagmap* m, int32_t n, const int32_t* ids, const double* sizes,
                         const double* rvec, const double* tvec, double* world_T_cam, int32_t* has_pose)
try {
  if (has_pose) *has_pose = 0;
  if (!m || n < 1 || !ids || !sizes || !rvec || !tvec) return -1;     // the reference only writes non-empty files (corner_detections.cpp:43)
  Frame f;
  f.ids.reserve(n); f.sizes.reserve(n); f.tag_T_cam.reserve(n);
  for (int i = 0; i < n; ++i) {
    f.ids.push_back(ids[i]);
    f.sizes.push_back(sizes[i]);
    f.tag_T_cam.push_back(rigid_inverse(from_rt(rvec + 3 * i, tvec + 3 * i)));   // camera_pose.cpp:172
  }
  // everything the rest of the call can append is reserved now: this frame's tags plus those of every deferred frame a
  // retry may localise.  A failed reservation throws BEFORE the map is touched; after it nothing below allocates.
  size_t may_add = (size_t)n;
  for (int32_t u : m->unreferenced) may_add += m->frames[u].ids.size();
  m->frames.reserve(m->frames.size() + 1);
  m->unreferenced.reserve(m->unreferenced.size() + 1);
  m->ids.reserve(m->ids.size() + may_add);
  m->sizes.reserve(m->sizes.size() + may_add);
  m->w_T_tag.reserve(m->w_T_tag.size() + may_add);
  m->frames.push_back(std::move(f));
  const int fn = (int)m->frames.size() - 1;
  int known = 0;
  const int status = m->reader(fn, known);                               // fileStream (:267-285)
  if (status == RCC_MAP_WORLD_PRES || status == RCC_MAP_KNOWN_TAG) {
    m->calc(fn, known);
    m->poll();
  } else {
    m->unreferenced.push_back(fn);
  }
  if (m->frames[fn].has_pose) {
    if (has_pose) *has_pose = 1;
    if (world_T_cam) memcpy(world_T_cam, m->frames[fn].w_T_cam.v, sizeof(double) * 16);
  }
  return status;
} catch (...) {
  if (has_pose) *has_pose = 0;
  return -1;
}

int rcc_tagmap_frame_pose(const rcc_tagmap* m, int32_t frame, double* world_T_cam)
{
  if (!m || frame < 0 || frame >= (int)m->frames.size() || !m->frames[frame].has_pose) return 0;
  if (world_T_cam) memcpy(world_T_cam, m->frames[frame].w_T_cam.v, sizeof(double) * 16);
  return 1;
}
int32_t rcc_tagmap_ntags(const rcc_tagmap* m) { return m ? (int32_t)m->ids.size() : 0; }
int rcc_tagmap_tag(const rcc_tagmap* m, int32_t i, int32_t* id, double* size, double* T)
{
  if (!m || i < 0 || i >= (int)m->ids.size()) return 0;
  if (id) *id = m->ids[i];
  if (size) *size = m->sizes[i];
  if (T) memcpy(T, m->w_T_tag[i].v, sizeof(double) * 16);
  return 1;
}
int32_t rcc_tagmap_pending(const rcc_tagmap* m) { return m ? (int32_t)m->unreferenced.size() : 0; }

// corner_detections.cpp:18-39 (yamlDump) + the trailing "\n" of :59
size_t rcc_yaml_detections(char* buf, size_t cap, int32_t n, const int32_t* ids, const double* sizes, const int32_t* corners)
try {
  if (n < 0 || (n > 0 && (!ids || !sizes || !corners))) return emit_nothing(buf, cap);
  std::string s = "detections:";
  for (int i = 0; i < n; ++i) {
    s += "\n - targetID: " + std::to_string(ids[i]);
    s += "\n   size: [ " + to6(sizes[i]) + ", " + to6(sizes[i]) + " ]";
    s += "\n   corners:";
    for (int k = 0; k < 4; ++k)
      s += "\n    " + std::to_string(k) + ": [ " + std::to_string(corners[(i * 4 + k) * 2]) + ", " + std::to_string(corners[(i * 4 + k) * 2 + 1]) + " ]";
  }
  s += "\n";
  return emit(s, buf, cap);
} catch (...) {
  return emit_nothing(buf, cap);
}

// camera_pose.cpp:83-100 (worldAppend)
size_t rcc_yaml_world_T_camera(char* buf, size_t cap, const double* T)
try {
  if (!T) return emit_nothing(buf, cap);
  M4 m; memcpy(m.v, T, sizeof(m.v));
  double r[3];
  rot_to_rvec(m, r);
  std::string s = "world_T_camera:";
  s += "\n rotation: [ " + to6(r[0]) + " , " + to6(r[1]) + " , " + to6(r[2]) + " ]";
  s += "\n translation: [ " + to6(T[3]) + " , " + to6(T[7]) + " , " + to6(T[11]) + " ]";
  return emit(s, buf, cap);
} catch (...) {
  return emit_nothing(buf, cap);
}

// camera_pose.cpp:103-129 (targetDump)
size_t rcc_yaml_targets(const rcc_tagmap* m, char* buf, size_t cap)
try {
  std::string s = "targets:";
  for (size_t i = 0; m && i < m->ids.size(); ++i) {
    double r[3];
    rot_to_rvec(m->w_T_tag[i], r);
    const double* T = m->w_T_tag[i].v;
    const double h = m->sizes[i] / 2;
    s += "\n - targetID: " + std::to_string(m->ids[i]);
    s += "\n   world_T_target:";
    s += "\n    rotation: [ " + to6(r[0]) + " , " + to6(r[1]) + " , " + to6(r[2]) + " ]";
    s += "\n    translation: [ " + to6(T[3]) + " , " + to6(T[7]) + " , " + to6(T[11]) + " ]";
    s += "\n   obj_points_in_target:";
    s += "\n    0: [ " + to6(-h) + ", " + to6(-h) + ", " + std::to_string(0) + " ]";
    s += "\n    1: [ " + to6(h) + ", " + to6(-h) + ", " + std::to_string(0) + " ]";
    s += "\n    2: [ " + to6(h) + ", " + to6(h) + ", " + std::to_string(0) + " ]";
    s += "\n    3: [ " + to6(-h) + ", " + to6(h) + ", " + std::to_string(0) + " ]";
  }
  return emit(s, buf, cap);
} catch (...) {
  return emit_nothing(buf, cap);
}

}  // extern "C"
