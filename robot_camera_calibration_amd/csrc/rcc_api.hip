// rcc_api.hip -- host side of the C ABI declared in include/rcc.h: handle life cycle, argument
// checking, stream-ordered launches of the stage kernels, result hand-over.
//
// Each entry point stands in for a reference interface (cited in include/rcc.h):
//   rcc_detect_batch     <- the "tag_detections" publisher consumed at corner_detections.cpp:41-56,78
//   rcc_solve_pnp_batch  <- cv::solvePnP(..., false, CV_ITERATIVE) at camera_pose.cpp:163
//   rcc_rodrigues_*      <- cv::Rodrigues at camera_pose.cpp:93,116,164 / opt_visualization.cpp:36
// Nothing here falls back to a CPU implementation: without a HIP device rcc_create fails with
// RCC_ERR_DEVICE.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <new>
#include <algorithm>
#include <vector>
#include "rcc_internal.h"
#include "../../include/rcc_debug.h"
#include "pnp_core.h"

#define HIPCHK(h, expr)                                                                   \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) {                                                               \
      if (h) snprintf((h)->err, sizeof((h)->err), "%s: %s", #expr, hipGetErrorString(e_)); \
      return RCC_ERR_DEVICE;                                                              \
    }                                                                                     \
  } while (0)

extern "C" {

int rcc_abi_version(void) { return RCC_ABI_VERSION; }

const char* rcc_status_string(int s)
{
  switch (s) {
    case RCC_OK: return "ok";
    case RCC_ERR_ARG: return "invalid argument";
    case RCC_ERR_UNSUPPORTED: return "unsupported configuration";
    case RCC_ERR_DEVICE: return "HIP device error";
    case RCC_ERR_CAPACITY: return "batch exceeds handle capacity";
    case RCC_ERR_NOMEM: return "out of memory";
    case RCC_ERR_STATE: return "call out of order (submit / collect)";
    default: return "unknown status";
  }
}

const char* rcc_last_device_error(const rcc_handle* h) { return h ? h->err : ""; }

void rcc_default_config(rcc_config* c)
{
  if (!c) return;
  memset(c, 0, sizeof(*c));
  c->struct_size = sizeof(rcc_config);
  c->abi_version = RCC_ABI_VERSION;
  c->width = 640; c->height = 480; c->stride_bytes = 640 * 3; c->pixfmt = RCC_PIX_BGR8;
  c->frame_bytes = (int64_t)640 * 3 * 480;
  c->K[0] = c->K[4] = 0.9 * 640; c->K[2] = (640 - 1) * 0.5; c->K[5] = (480 - 1) * 0.5; c->K[8] = 1.0;
  c->dist_model = RCC_DIST_PLUMB_BOB;
  c->undistort = 1;
  c->D[0] = -0.28; c->D[1] = 0.07; c->D[2] = 2e-4; c->D[3] = -1e-4; c->D[4] = 0.0;
  // (16, 10240): the pair keeps the flat-tile skip exact (rcc_dense_allow_skip) and finds the board under blur up to sigma 2 px and
  // 60 % shading (profiles/r04_b_optics_table.json); rounds 1-3 shipped (32, 200000), tuned on razor-edged renders only
  c->thr_min_contrast = 16;
  c->harris_thresh = 10240;
  c->cand_margin = 8;
  c->max_candidates = 2048;
  c->nms_radius = 5;
  c->xj_check = 1;
  c->max_kept = 256;
  c->subpix_win = 5;
  c->subpix_max_iter = 30;
  c->subpix_eps = 1e-3;
  c->target_kind = RCC_TARGET_CHECKERBOARD;
  c->board_cols = 8; c->board_rows = 6; c->board_square = 0.108; c->board_id = 0;
  c->max_targets = 1;
  c->reference_mode = 0;
  c->pnp_use_mfma = 0;
  c->device = 0;
  c->batch_capacity = 16;
}

static int validate(const rcc_config* c)
{
  if (!c) return RCC_ERR_ARG;
  if (c->struct_size != sizeof(rcc_config) || c->abi_version != RCC_ABI_VERSION) return RCC_ERR_ARG;
  if (c->width < 1 || c->height < 1 || c->width > 16384 || c->height > 16384) return RCC_ERR_ARG;
  if (c->pixfmt != RCC_PIX_MONO8 && c->pixfmt != RCC_PIX_BGR8 && c->pixfmt != RCC_PIX_RGB8) return RCC_ERR_ARG;
  const int ch = c->pixfmt == RCC_PIX_MONO8 ? 1 : 3;
  if (c->stride_bytes < c->width * ch) return RCC_ERR_ARG;
  if (c->frame_bytes < (int64_t)c->stride_bytes * c->height) return RCC_ERR_ARG;
  if (c->dist_model < RCC_DIST_NONE || c->dist_model > RCC_DIST_FISHEYE) return RCC_ERR_ARG;
  if (!(c->K[0] > 0.0) || !(c->K[4] > 0.0)) return RCC_ERR_ARG;
  if (c->max_candidates < 1 || c->max_candidates > 4096) return RCC_ERR_ARG;
  if (c->max_kept < 1 || (c->target_kind != RCC_TARGET_FIDUCIAL && c->max_kept > RCC_MAX_KEPT)) return RCC_ERR_ARG;
  if (c->subpix_win < 1 || c->subpix_win > 7 || c->subpix_max_iter < 1) return RCC_ERR_ARG;
  if (c->nms_radius < 0 || c->cand_margin < 0) return RCC_ERR_ARG;
  if (c->batch_capacity < 1) return RCC_ERR_ARG;
  if (c->max_targets < 1) return RCC_ERR_ARG;
  if (c->target_kind == RCC_TARGET_CHECKERBOARD) {
    if (c->board_cols < 2 || c->board_rows < 2 || c->board_cols > 16 || c->board_rows > 16) return RCC_ERR_ARG;
    if (c->board_cols * c->board_rows > RCC_MAX_BOARD_CORNERS || !(c->board_square > 0.0)) return RCC_ERR_ARG;
  } else if (c->target_kind == RCC_TARGET_FIDUCIAL) {
    if (!c->family_codes || c->family_n < 1 || c->family_n > 65536 || !(c->tag_size > 0.0)) return RCC_ERR_ARG;
    if (c->tag_max_hamming < 0 || c->tag_max_hamming > 8 || c->max_targets > 4096) return RCC_ERR_ARG;
    if (c->max_kept > RCC_MAX_KEPT_FIDUCIAL) return RCC_ERR_ARG;
    if (c->tag_refine != RCC_TAG_REFINE_EDGES && c->tag_refine != RCC_TAG_REFINE_CORNER_SUBPIX) return RCC_ERR_ARG;
  } else {
    return RCC_ERR_ARG;
  }
  // solvePnP has no fisheye model (the reference only has the 5-coefficient plumb-bob,
  // camera_pose.cpp:39): fisheye frames must be undistorted first
  if (c->dist_model == RCC_DIST_FISHEYE && !c->undistort) return RCC_ERR_UNSUPPORTED;
  return RCC_OK;
}

void rcc_destroy(rcc_handle* h)
{
  if (!h) return;
  if (h->tail_stream) { (void)hipStreamSynchronize(h->tail_stream); (void)hipStreamDestroy(h->tail_stream); }
  for (int k = 0; k < 2; ++k) { if (h->tail_val[k]) (void)hipEventDestroy(h->tail_val[k]); if (h->tail_mark[k]) (void)hipEventDestroy(h->tail_mark[k]); if (h->tail_done[k]) (void)hipEventDestroy(h->tail_done[k]); }
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  void* ptrs[] = { h->d_map, h->d_tilebox, h->d_flat, h->d_thr, h->d_grey, h->d_bin, h->d_cand, h->d_cand_count, h->d_pre, h->d_npre, h->d_pre_xy,
                   h->d_kept, h->d_kept_xy, h->d_ref_xy, h->d_fc, h->d_det, h->d_ndet, h->d_stage, h->d_pnp_buf,
                   h->d_board_obj, h->d_img_scratch, h->d_family, h->d_sp_tab, h->d_synth_tmp };
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (h->h_det) (void)hipHostFree(h->h_det);
  if (h->h_ndet) (void)hipHostFree(h->h_ndet);
  if (h->h_det2) (void)hipHostFree(h->h_det2);
  if (h->h_ndet2) (void)hipHostFree(h->h_ndet2);
  for (auto& p : h->h_fc) if (p) (void)hipHostFree(p);
  for (auto& ps : h->pstream) if (ps) (void)hipStreamSynchronize(ps);
  if (h->fc_stream) { (void)hipStreamSynchronize(h->fc_stream); (void)hipStreamDestroy(h->fc_stream); }
  for (auto& e : h->sub_ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : h->fc_ready) if (e) (void)hipEventDestroy(e);
  for (auto& e : h->fc_done) if (e) (void)hipEventDestroy(e);
  for (auto& p : h->sub_t_ev) for (auto& e : p) if (e) (void)hipEventDestroy(e);
  for (auto& e : h->ev) if (e) (void)hipEventDestroy(e);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  for (auto& ps : h->pstream) if (ps) (void)hipStreamDestroy(ps);
  for (auto& e : h->pev) if (e) (void)hipEventDestroy(e);
  for (auto& e : h->cev) if (e) (void)hipEventDestroy(e);
  delete h;
}

int rcc_create(const rcc_config* cfg, rcc_handle** out)
{
  if (!out) return RCC_ERR_ARG;
  *out = nullptr;
  int v = validate(cfg);
  if (v != RCC_OK) return v;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || cfg->device < 0 || cfg->device >= ndev) return RCC_ERR_DEVICE;
  rcc_handle* h = new (std::nothrow) rcc_handle();
  if (!h) return RCC_ERR_NOMEM;
  memset(h, 0, sizeof(*h));
  h->cfg = *cfg;
  h->device = cfg->device;
  h->undist = cfg->undistort && cfg->dist_model != RCC_DIST_NONE;
  h->dense_variant = -1;
  h->fuse_grid_pnp = 1;
  h->pipeline_chunks = 1;   // measured: chunking the batch over two streams is slower at every chunk count
  h->ingest_variant = -1;
  h->dense_skip = 1;
  h->pnp_variant = -1;
  h->kept_cap = RCC_MAX_KEPT_FIDUCIAL;      // capacity (and stride) of the per-frame lists after suppression, whatever the target: the
                                            // board path's cfg.max_kept (<= 256) bounds the VALIDATED list only (round 4: cluttered scenes)
  h->pnp_solver = 1;
  h->pnp_use_mfma = cfg->pnp_use_mfma ? 1 : 0;
  h->ingest_table = 1;
#ifdef RCC_EXPERIMENTS
  if (const char* e = getenv("RCC_PNP_SOLVER")) h->pnp_solver = atoi(e);
#endif
  h->sp.win = cfg->subpix_win;
  h->sp.max_iter = cfg->subpix_max_iter;
  h->sp.eps2 = cfg->subpix_eps * cfg->subpix_eps;
  if (cfg->target_kind == RCC_TARGET_FIDUCIAL && cfg->tag_refine == RCC_TAG_REFINE_EDGES) {
    // refine_edges form: the a5 pass in front of the quad search is a coarse localisation (include/rcc.h)
    if (h->sp.max_iter > RCC_TAG_COARSE_ITERS) h->sp.max_iter = RCC_TAG_COARSE_ITERS;
    const double eps = cfg->subpix_eps > RCC_TAG_COARSE_EPS ? cfg->subpix_eps : RCC_TAG_COARSE_EPS;
    h->sp.eps2 = eps * eps;
  }
  for (int k = -cfg->subpix_win; k <= cfg->subpix_win; ++k) {
    double t = (double)k / (double)cfg->subpix_win;
    h->sp.m1[k + cfg->subpix_win] = exp(-(t * t));   // host libm, as the specification does
  }
  const size_t B = (size_t)cfg->batch_capacity, px = (size_t)cfg->width * cfg->height;
#define ALLOC(ptr, bytes)                                                                           \
  do {                                                                                              \
    hipError_t e_ = hipMalloc((void**)&(ptr), (bytes));                                             \
    if (e_ != hipSuccess) { rcc_destroy(h); return e_ == hipErrorOutOfMemory ? RCC_ERR_NOMEM : RCC_ERR_DEVICE; } \
  } while (0)
  if (hipSetDevice(h->device) != hipSuccess) { delete h; return RCC_ERR_DEVICE; }
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return RCC_ERR_DEVICE; }
  for (auto& e : h->ev) if (hipEventCreate(&e) != hipSuccess) { rcc_destroy(h); return RCC_ERR_DEVICE; }
  for (auto& ps : h->pstream) if (hipStreamCreateWithFlags(&ps, hipStreamNonBlocking) != hipSuccess) { rcc_destroy(h); return RCC_ERR_DEVICE; }
  if (hipStreamCreateWithFlags(&h->fc_stream, hipStreamNonBlocking) != hipSuccess) { rcc_destroy(h); return RCC_ERR_DEVICE; }
  for (auto& e : h->pev) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { rcc_destroy(h); return RCC_ERR_DEVICE; }
  ALLOC(h->d_grey, B * px);
  // d_bin (the full {0,127,255} image, B * px bytes) is allocated on first need (ensure_bin): the default detect path keeps
  // the binary image as the compact threshold map d_thr and never touches it
  ALLOC(h->d_thr, B * (size_t)((cfg->width + RCC_BAND_W - 1) / RCC_BAND_W) * (size_t)((cfg->height + 3) >> 2) * RCC_THR_PITCH);
  ALLOC(h->d_cand, B * (size_t)cfg->max_candidates * sizeof(rcc_cand));
  ALLOC(h->d_cand_count, B * sizeof(int32_t));
  ALLOC(h->d_pre, B * (size_t)h->kept_cap * sizeof(rcc_cand));
  ALLOC(h->d_npre, B * sizeof(int32_t));
  ALLOC(h->d_pre_xy, B * (size_t)h->kept_cap * 2 * sizeof(double));
  ALLOC(h->d_kept, B * RCC_MAX_KEPT * sizeof(rcc_cand));
  ALLOC(h->d_kept_xy, B * RCC_MAX_KEPT * 2 * sizeof(double));
  ALLOC(h->d_fc, B * sizeof(rcc_frame_corners));
  ALLOC(h->d_det, B * (size_t)cfg->max_targets * sizeof(rcc_detection));
  ALLOC(h->d_ndet, B * sizeof(int32_t));
  ALLOC(h->d_img_scratch, B * 2 * RCC_MAX_BOARD_CORNERS * sizeof(double));
  ALLOC(h->d_board_obj, 3 * RCC_MAX_BOARD_CORNERS * sizeof(double));
  if (hipHostMalloc((void**)&h->h_det, B * (size_t)cfg->max_targets * sizeof(rcc_detection)) != hipSuccess ||
      hipHostMalloc((void**)&h->h_ndet, B * sizeof(int32_t)) != hipSuccess ||
      hipHostMalloc((void**)&h->h_det2, B * (size_t)cfg->max_targets * sizeof(rcc_detection)) != hipSuccess ||
      hipHostMalloc((void**)&h->h_ndet2, B * sizeof(int32_t)) != hipSuccess) { rcc_destroy(h); return RCC_ERR_NOMEM; }
  for (auto& e : h->sub_ev) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { rcc_destroy(h); return RCC_ERR_DEVICE; }
  for (auto& e : h->fc_ready) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { rcc_destroy(h); return RCC_ERR_DEVICE; }
  for (auto& e : h->fc_done) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { rcc_destroy(h); return RCC_ERR_DEVICE; }
  for (auto& p : h->sub_t_ev) for (auto& e : p) if (hipEventCreate(&e) != hipSuccess) { rcc_destroy(h); return RCC_ERR_DEVICE; }
  for (float& m : h->last_step_ms) m = -1.0f;
  {
    // object points of the board: index = row*cols + col, x right, y up, z = 0, origin at the
    // centre -- the object-frame convention of camera_pose.cpp:158-161
    std::vector<double> obj(3 * RCC_MAX_BOARD_CORNERS, 0.0);
    for (int r = 0; r < cfg->board_rows; ++r)
      for (int c = 0; c < cfg->board_cols; ++c) {
        double* o = &obj[3 * (r * cfg->board_cols + c)];
        o[0] = ((double)c - 0.5 * (double)(cfg->board_cols - 1)) * cfg->board_square;
        o[1] = (0.5 * (double)(cfg->board_rows - 1) - (double)r) * cfg->board_square;
        o[2] = 0.0;
      }
    if (hipMemcpy(h->d_board_obj, obj.data(), obj.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
      rcc_destroy(h);
      return RCC_ERR_DEVICE;
    }
  }
  {
    // sub-pixel sample tables, one entry per lane (k_subpix.hip)
    const int win = cfg->subpix_win, ww = 2 * win + 1, pw = 2 * win + 3;
    std::vector<rcc_subpix_lane> tab(64);
    for (int lane = 0; lane < 64; ++lane) {
      rcc_subpix_lane& e = tab[lane];
      memset(&e, 0, sizeof(e));
      for (int t = 0; t < RCC_SP_PT; ++t) {
        const int idx = lane + 64 * t, i = idx / pw, j = idx - i * pw;
        e.poff[t] = (i - win - 1) * cfg->width + (j - win - 1);
      }
      for (int t = 0; t < RCC_SP_GT; ++t) {
        const int k = lane + 64 * t, i = k / ww, j = k - i * ww;
        e.goff[t] = (i + 1) * pw + (j + 1);
        const bool ok = k < ww * ww;
        e.gm[t] = ok ? h->sp.m1[i] * h->sp.m1[j] : 0.0;
        e.gpx[t] = (int8_t)(ok ? j - win : 0); e.gpy[t] = (int8_t)(ok ? i - win : 0);
      }
    }
    ALLOC(h->d_sp_tab, tab.size() * sizeof(rcc_subpix_lane));
    if (hipMemcpy(h->d_sp_tab, tab.data(), tab.size() * sizeof(rcc_subpix_lane), hipMemcpyHostToDevice) != hipSuccess) {
      rcc_destroy(h);
      return RCC_ERR_DEVICE;
    }
  }
  if (cfg->target_kind == RCC_TARGET_FIDUCIAL) {
    ALLOC(h->d_ref_xy, B * (size_t)h->kept_cap * 2 * sizeof(double));
    ALLOC(h->d_family, (size_t)cfg->family_n * sizeof(uint64_t));
    if (hipMemcpy(h->d_family, cfg->family_codes, (size_t)cfg->family_n * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess) {
      rcc_destroy(h);
      return RCC_ERR_DEVICE;
    }
    h->cfg.family_codes = nullptr;   // the caller's table is not referenced after create
  }
  *out = h;
  return RCC_OK;
}

// the handle's own full binary image: needed when a batch's binary image is not left as the compact threshold map
static int ensure_bin(rcc_handle* h)
{
  if (h->d_bin) return RCC_OK;
  const size_t bytes = (size_t)h->cfg.batch_capacity * h->cfg.width * h->cfg.height;
  hipError_t e = hipMalloc((void**)&h->d_bin, bytes);
  if (e != hipSuccess) { h->d_bin = nullptr; return e == hipErrorOutOfMemory ? RCC_ERR_NOMEM : RCC_ERR_DEVICE; }
  return RCC_OK;
}
// will rcc_launch_dense (with want_thr set) write d_thr instead of a full image?  (mirrors its variant choice)
static bool detect_needs_bin(const rcc_handle* h)
{
  const bool band_variant = h->dense_variant < 0 || h->dense_variant == 1 || h->dense_variant == 3 || h->dense_variant == 4;   // (3 runs as 1 in the product library)
  return h->keep_bin || !band_variant || !rcc_dense_band_supported(h, h->d_grey, nullptr);
}

int rcc_set_dense_variant(rcc_handle* h, int variant)
{
  if (!h) return RCC_ERR_ARG;
  int p = h->dense_variant;
  h->dense_variant = variant;
  return p;
}
#ifdef RCC_EXPERIMENTS
// k_dense_wave as gangs of eight windows: sync_rows = 0 off, else a power of two (tile rows between the gang's barriers);
// segments = 0: as the single-window form
int rcc_set_dense_gang(rcc_handle* h, int sync_rows, int segments)
{
  if (!h || sync_rows < 0 || (sync_rows & (sync_rows - 1)) || segments < 0) return RCC_ERR_ARG;
  int p = h->dense_gang_sync;
  h->dense_gang_sync = sync_rows;
  h->dense_gang_seg = segments;
  return p;
}
#endif
int rcc_set_dense_skip(rcc_handle* h, int on)
{
  if (!h) return RCC_ERR_ARG;
  int p = h->dense_skip;
  h->dense_skip = on ? 1 : 0;
  return p;
}
int rcc_set_pnp_variant(rcc_handle* h, int variant)
{
  if (!h) return RCC_ERR_ARG;
  int p = h->pnp_variant;
  h->pnp_variant = variant;
  return p;
}
int rcc_set_pnp_mfma(rcc_handle* h, int on)
{
  if (!h) return RCC_ERR_ARG;
  int p = h->pnp_use_mfma;
  h->pnp_use_mfma = on ? 1 : 0;
  return p;
}
// variant: 0 gather, 1 staged with the tabulated map (default where the geometry allows; 128 x 16 destination tiles), 2 staged
// recomputing the map per block (the form a handle falls back to when the table cannot be allocated), 3 staged with the tabulated
// map and the 128 x 8 tiles of rounds 1-3 (experiments library only; the product library runs 1 in its place); -1 automatic
int rcc_set_ingest_variant(rcc_handle* h, int variant)
{
  if (!h) return RCC_ERR_ARG;
  int p = h->ingest_variant;
  if (p == 1 && !h->ingest_table) p = 2;
  else if (p == 1 && h->ingest_tile8) p = 3;
  h->ingest_table = (variant == 2) ? 0 : 1;
  h->ingest_tile8 = (variant == 3) ? 1 : 0;
  h->ingest_variant = (variant == 2 || variant == 3) ? 1 : variant;
  return p;
}

const char* rcc_last_dense_kernel(const rcc_handle* h) { return (h && h->dense_kernel) ? h->dense_kernel : ""; }

int rcc_last_timings(const rcc_handle* h, float* ms, int32_t n)
{
  if (!h || !ms) return RCC_ERR_ARG;
  int k = n < 5 ? n : 5;
  for (int i = 0; i < k; ++i) ms[i] = h->last_ms[i];
  return k;
}

int rcc_last_step_times(const rcc_handle* h, float* ms, int32_t n)
{
  if (!h || !ms) return RCC_ERR_ARG;
  int k = n < 7 ? n : 7;
  for (int i = 0; i < k; ++i) ms[i] = h->last_step_ms[i];
  return k;
}

static bool records_fit(const rcc_handle* h, int nframes);

// ---- stages ------------------------------------------------------------------------------------
int rcc_stage_ingest(rcc_handle* h, const void* d_frames, int32_t nframes, void* d_grey, void* stream)
{
  if (!h || !d_frames || !d_grey || nframes < 0) return RCC_ERR_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  HIPCHK(h, rcc_launch_ingest(h, (const uint8_t*)d_frames, nframes, (uint8_t*)d_grey, s));
  HIPCHK(h, hipStreamSynchronize(s));
  return RCC_OK;
}

int rcc_stage_threshold_corner(rcc_handle* h, const void* d_grey, int32_t nframes, void* d_bin,
                               void* d_cand, void* d_cand_count, void* stream)
{
  if (!h || !d_grey || !d_bin || !d_cand || !d_cand_count || nframes < 0) return RCC_ERR_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  h->want_thr = 0;          // this entry point hands the full binary image to the caller
  HIPCHK(h, rcc_launch_dense(h, (const uint8_t*)d_grey, nframes, (uint8_t*)d_bin, (rcc_cand*)d_cand, (int32_t*)d_cand_count, s));
  HIPCHK(h, hipStreamSynchronize(s));
  return RCC_OK;
}

int rcc_time_dense(rcc_handle* h, const void* d_grey, int32_t nframes, void* d_bin, void* d_cand,
                   void* d_cand_count, int32_t reps, float* mean_ms)
{
  if (!h || !d_grey || !d_cand || !d_cand_count || nframes < 1 || reps < 1 || !mean_ms) return RCC_ERR_ARG;
  if (!d_bin && nframes > h->cfg.batch_capacity) return RCC_ERR_CAPACITY;
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t s = h->stream;
  // the count reset (a tiny memset) is part of every launch of the pass; it stays inside
  h->want_thr = d_bin ? 0 : 1;      // d_bin == NULL: the form rcc_detect_batch runs (compact threshold map in the handle)
  if (!d_bin) {
    if (detect_needs_bin(h)) { int r = ensure_bin(h); if (r != RCC_OK) return r; }
    d_bin = h->d_bin;
  }
  HIPCHK(h, hipEventRecord(h->ev[6], s));
  for (int r = 0; r < reps; ++r)
    HIPCHK(h, rcc_launch_dense(h, (const uint8_t*)d_grey, nframes, (uint8_t*)d_bin, (rcc_cand*)d_cand, (int32_t*)d_cand_count, s));
  HIPCHK(h, hipEventRecord(h->ev[7], s));
  HIPCHK(h, hipEventSynchronize(h->ev[7]));
  float ms = 0.f;
  HIPCHK(h, hipEventElapsedTime(&ms, h->ev[6], h->ev[7]));
  *mean_ms = ms / reps;
  return RCC_OK;
}

int rcc_time_ingest(rcc_handle* h, const void* d_frames, int32_t nframes, void* d_grey, int32_t reps, float* mean_ms)
{
  if (!h || !d_frames || !d_grey || nframes < 1 || reps < 1 || !mean_ms) return RCC_ERR_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t s = h->stream;
  HIPCHK(h, hipEventRecord(h->ev[6], s));
  for (int r = 0; r < reps; ++r) HIPCHK(h, rcc_launch_ingest(h, (const uint8_t*)d_frames, nframes, (uint8_t*)d_grey, s));
  HIPCHK(h, hipEventRecord(h->ev[7], s));
  HIPCHK(h, hipEventSynchronize(h->ev[7]));
  float ms = 0.f;
  HIPCHK(h, hipEventElapsedTime(&ms, h->ev[6], h->ev[7]));
  *mean_ms = ms / reps;
  return RCC_OK;
}

// list + subpix + validate/grid + pnp, then results to the host.  Events: ev[2]..ev[5].
// list -> sub-pixel -> target identification -> pose, for the frames the (possibly offset) handle view covers
// tev: three timing events (recorded in front of the list stage, behind the sub-pixel / quad stage, behind the pose stage) or NULL
static int launch_targets(rcc_handle* h, const uint8_t* d_grey, const uint8_t* d_bin, const rcc_cand* d_cand,
                          const int32_t* d_cand_count, int nframes, hipStream_t s, hipEvent_t* tev)
{
  const bool fid = h->cfg.target_kind == RCC_TARGET_FIDUCIAL;
  const bool timed = tev != nullptr;
  if (timed) HIPCHK(h, hipEventRecord(tev[0], s));
  HIPCHK(h, rcc_launch_list(h, d_cand, d_cand_count, nframes, s));
  HIPCHK(h, rcc_launch_subpix(h, d_grey, nframes, s));
  const bool fused = !fid && h->fuse_grid_pnp && rcc_grid_pnp_applicable(h);   // lattice indexing + pose in one launch
  if (fid) HIPCHK(h, rcc_launch_fid(h, d_grey, nframes, s));
  else if (!fused) HIPCHK(h, rcc_launch_grid(h, d_grey, d_bin, nframes, s));
  if (timed) HIPCHK(h, hipEventRecord(tev[1], s));
  if (fid) HIPCHK(h, rcc_launch_pnp_tags(h, nframes, s));
  else if (fused) HIPCHK(h, rcc_launch_grid_pnp(h, d_grey, d_bin, nframes, s));
  else HIPCHK(h, rcc_launch_pnp_board(h, nframes, s));
  if (timed) HIPCHK(h, hipEventRecord(tev[2], s));
  return RCC_OK;
}

// The previous submission's corner tables may still be on their way to the host (copy stream): the list stage is the first kernel
// that writes d_fc again, so the stream that is about to launch it waits for that copy (a device-side wait, long satisfied by then).
static int wait_corner_copy(rcc_handle* h, hipStream_t s)
{
  if (h->fc_pending) {
    HIPCHK(h, hipStreamWaitEvent(s, h->fc_done[h->fc_pending - 1], 0));
    h->fc_pending = 0;
  }
  return RCC_OK;
}

// device -> host copy of the records, synchronise, compact into the caller's array
static int collect_targets(rcc_handle* h, int nframes, rcc_detection* det, int32_t* ndet, rcc_frame_corners* corners,
                           hipStream_t s, bool timed)
{
  const bool fid = h->cfg.target_kind == RCC_TARGET_FIDUCIAL;
  const int slots = h->cfg.max_targets;
  HIPCHK(h, hipMemcpyAsync(h->h_det, h->d_det, sizeof(rcc_detection) * (size_t)nframes * (fid ? slots : 1), hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipMemcpyAsync(h->h_ndet, h->d_ndet, sizeof(int32_t) * (size_t)nframes, hipMemcpyDeviceToHost, s));
  if (corners) HIPCHK(h, hipMemcpyAsync(corners, h->d_fc, sizeof(rcc_frame_corners) * (size_t)nframes, hipMemcpyDeviceToHost, s));
  if (timed) HIPCHK(h, hipEventRecord(h->ev[5], s));
  HIPCHK(h, hipStreamSynchronize(s));
  int n = 0;
  for (int f = 0; f < nframes; ++f) {
    const int k = h->h_ndet[f] < slots ? h->h_ndet[f] : slots;
    for (int q = 0; q < k; ++q) {
      if (det) { det[n] = h->h_det[(size_t)f * (fid ? slots : 1) + q]; det[n].frame = f; }   // batch index (kernels number within their chunk)
      ++n;
    }
  }
  if (ndet) *ndet = n;
  if (timed) {
    (void)hipEventElapsedTime(&h->last_ms[2], h->ev[2], h->ev[3]);
    (void)hipEventElapsedTime(&h->last_ms[3], h->ev[3], h->ev[4]);
    (void)hipEventElapsedTime(&h->last_ms[4], h->ev[4], h->ev[5]);
  }
  return RCC_OK;
}

static int run_targets(rcc_handle* h, const uint8_t* d_grey, const uint8_t* d_bin, const rcc_cand* d_cand,
                       const int32_t* d_cand_count, int nframes, rcc_detection* det, int32_t* ndet,
                       rcc_frame_corners* corners, hipStream_t s)
{
  int r = launch_targets(h, d_grey, d_bin, d_cand, d_cand_count, nframes, s, &h->ev[2]);
  if (r != RCC_OK) return r;
  return collect_targets(h, nframes, det, ndet, corners, s, true);
}

// a copy of the handle whose per-frame buffers start at frame f0: the launchers index frames from 0
static rcc_handle handle_view(const rcc_handle* h, int f0)
{
  rcc_handle v = *h;
  const size_t o = (size_t)f0, px = (size_t)h->cfg.width * h->cfg.height;
  v.d_grey += o * px;
  if (v.d_bin) v.d_bin += o * px;
  v.d_thr += o * (size_t)((h->cfg.width + RCC_BAND_W - 1) / RCC_BAND_W) * (size_t)((h->cfg.height + 3) >> 2) * RCC_THR_PITCH;
  v.d_cand += o * (size_t)h->cfg.max_candidates; v.d_cand_count += o;
  v.d_pre += o * (size_t)h->kept_cap; v.d_npre += o; v.d_pre_xy += o * (size_t)h->kept_cap * 2;
  if (v.d_ref_xy) v.d_ref_xy += o * (size_t)h->kept_cap * 2;
  v.d_kept += o * RCC_MAX_KEPT; v.d_kept_xy += o * RCC_MAX_KEPT * 2;
  v.d_fc += o; v.d_det += o * (size_t)h->cfg.max_targets; v.d_ndet += o;
  v.d_img_scratch += o * 2 * RCC_MAX_BOARD_CORNERS;
  return v;
}

int rcc_stage_targets(rcc_handle* h, const void* d_grey, const void* d_bin, const void* d_cand,
                      const void* d_cand_count, int32_t nframes, rcc_detection* det, int32_t* ndet,
                      rcc_frame_corners* corners, void* stream)
{
  if (!h || !d_grey || !d_bin || !d_cand || !d_cand_count || nframes < 0) return RCC_ERR_ARG;
  if (nframes > h->cfg.batch_capacity) return RCC_ERR_CAPACITY;
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  h->bin_from_thr = 0;      // the caller's binary image is the source
  return run_targets(h, (const uint8_t*)d_grey, (const uint8_t*)d_bin, (const rcc_cand*)d_cand,
                     (const int32_t*)d_cand_count, nframes, det, ndet, corners, s);
}

// the staging buffer for host-resident input (sized for batch_capacity once)
static int ensure_stage(rcc_handle* h, size_t need, bool may_realloc)
{
  if (need <= h->stage_bytes) return RCC_OK;
  // a staging buffer that exists cannot be replaced under a batch in flight (it cannot be too small either: it is allocated
  // for the handle's capacity); one that does not exist yet -- the batch in flight came from device memory -- is simply made
  if (!may_realloc && h->d_stage) return RCC_ERR_STATE;
  if (h->d_stage) (void)hipFree(h->d_stage);
  h->d_stage = nullptr;
  h->stage_bytes = 0;
  const size_t cap = (size_t)h->cfg.frame_bytes * h->cfg.batch_capacity;
  if (hipMalloc((void**)&h->d_stage, cap) != hipSuccess) return RCC_ERR_NOMEM;
  h->stage_bytes = cap;
  return RCC_OK;
}

// Frames per chunk of the host-input pipeline, or 0 when the batch goes over in one copy.  The copy engine is fastest on
// transfers of 64-256 MiB (one 6-GB copy: 50-55 GB/s; chunks: 56-57 GB/s, scratch/t_h2d.py), and every chunk's kernels
// run under the following chunks' copies, so only the last chunk's kernels are exposed: chunks of about 192 MiB, at
// most RCC_HOST_CHUNKS of them, and at least 8 frames each (the tail stages are a fixed-length chain per launch).
static int host_chunk_frames(const rcc_handle* h, int nframes)
{
  if (h->host_chunk_frames < 0) return 0;
  long long per = h->host_chunk_frames > 0 ? h->host_chunk_frames : (long long)((192ull << 20) / (unsigned long long)h->cfg.frame_bytes);
  if (per < 8) per = 8;
  if ((nframes + per - 1) / per > RCC_HOST_CHUNKS) per = (nframes + RCC_HOST_CHUNKS - 1) / RCC_HOST_CHUNKS;
  return nframes >= 2 * per ? (int)per : 0;
}

// Host-resident batch as a pipeline: chunk c is copied on one of two copy streams (alternating: the next copy is queued
// while this one runs) and its kernels -- ingest, threshold + corner pass, targets, over the chunk's slices of the
// handle's buffers -- follow on the caller's stream s as soon as the copy has landed, under the copies of the chunks
// behind it.  Same records as the one-copy form (chunks only touch their own frames' slices).
static int launch_host_pipeline(rcc_handle* h, const uint8_t* host_frames, int nframes, int per, hipStream_t s)
{
  const size_t fb = (size_t)h->cfg.frame_bytes;
  const int nchunks = (nframes + per - 1) / per;
  { int rw = wait_corner_copy(h, s); if (rw != RCC_OK) return rw; }
  // the staging buffer and the per-frame buffers are this handle's only set: the copies wait for everything queued on s
  HIPCHK(h, hipEventRecord(h->pev[0], s));
  for (int k = 0; k < 2; ++k) HIPCHK(h, hipStreamWaitEvent(h->pstream[k], h->pev[0], 0));
  // Issue order: copy 0, then per chunk c its kernels (they wait for copy c's event) FOLLOWED by copy c + 1.  With pinned input
  // every call returns at once and the order does not matter; with PAGEABLE input hipMemcpyAsync holds the host until the bytes
  // have gone through the runtime's bounce buffers -- issued in this order that wait falls while the device runs chunk c's
  // kernels, so copy and compute still overlap (all copies first, as rounds 1-3 queued them, would have serialised: no kernel
  // launched before the last copy had finished).
  auto issue_copy = [&](int c) -> int {
    const int f0 = c * per, f1 = (f0 + per < nframes) ? f0 + per : nframes;
    if (!h->cev[c]) HIPCHK(h, hipEventCreateWithFlags(&h->cev[c], hipEventDisableTiming));
    hipStream_t cs = h->pstream[c & 1];
    HIPCHK(h, hipMemcpyAsync(h->d_stage + (size_t)f0 * fb, host_frames + (size_t)f0 * fb, (size_t)(f1 - f0) * fb, hipMemcpyHostToDevice, cs));
    HIPCHK(h, hipEventRecord(h->cev[c], cs));
    return RCC_OK;
  };
  { int rc = issue_copy(0); if (rc != RCC_OK) return rc; }
  for (int c = 0; c < nchunks; ++c) {
    const int f0 = c * per, f1 = (f0 + per < nframes) ? f0 + per : nframes;
    HIPCHK(h, hipStreamWaitEvent(s, h->cev[c], 0));
    rcc_handle v = handle_view(h, f0);
    v.want_thr = h->keep_bin ? 0 : 1;
    if (v.dense_variant == 3) v.dense_variant = 1;      // (one flat-mask buffer per handle: see the chunked device path)
    hipError_t e = rcc_launch_ingest(&v, h->d_stage + (size_t)f0 * fb, f1 - f0, v.d_grey, s);
    h->d_map = v.d_map; h->d_tilebox = v.d_tilebox; h->map_failed = v.map_failed; h->map_th = v.map_th;     // tables built by the first launch belong to the handle
    if (e == hipSuccess) e = rcc_launch_dense(&v, v.d_grey, f1 - f0, v.d_bin, v.d_cand, v.d_cand_count, s);
    if (e != hipSuccess) { snprintf(h->err, sizeof(h->err), "%s", hipGetErrorString(e)); return RCC_ERR_DEVICE; }
    h->dense_kernel = v.dense_kernel;
    int r = launch_targets(&v, v.d_grey, v.d_bin, v.d_cand, v.d_cand_count, f1 - f0, s, nullptr);
    if (r != RCC_OK) { snprintf(h->err, sizeof(h->err), "%s", v.err); return r; }
    h->bin_from_thr = v.bin_from_thr;
    if (c + 1 < nchunks) { int rc = issue_copy(c + 1); if (rc != RCC_OK) return rc; }
  }
  for (float& m : h->last_ms) m = -1.0f;               // stages of different chunks overlap the copies: no separate durations
  return RCC_OK;
}

int rcc_detect_batch(rcc_handle* h, const void* frames, int32_t nframes, int32_t frames_mem,
                     rcc_detection* det, int32_t* ndet, rcc_frame_corners* corners, void* stream)
{
  if (!h || (!frames && nframes > 0) || nframes < 0) return RCC_ERR_ARG;
  if (frames_mem != RCC_MEM_HOST && frames_mem != RCC_MEM_DEVICE) return RCC_ERR_ARG;
  if (nframes > h->cfg.batch_capacity || !records_fit(h, nframes)) return RCC_ERR_CAPACITY;
  if (ndet) *ndet = 0;
  if (nframes == 0) return RCC_OK;
  if (h->sub_head != h->sub_tail) return RCC_ERR_STATE;              // submissions outstanding: collect them first
  HIPCHK(h, hipSetDevice(h->device));
  if (detect_needs_bin(h)) { int rb = ensure_bin(h); if (rb != RCC_OK) return rb; }
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  const uint8_t* d_frames = (const uint8_t*)frames;
  if (frames_mem == RCC_MEM_HOST) {
    { int rs = ensure_stage(h, (size_t)h->cfg.frame_bytes * nframes, true); if (rs != RCC_OK) return rs; }
    const int per = host_chunk_frames(h, nframes);
    if (per > 0) {
      int r = launch_host_pipeline(h, (const uint8_t*)frames, nframes, per, s);
      if (r != RCC_OK) return r;
      if (h->rec_table[0]) HIPCHK(h, rcc_launch_pack_records(h, nframes, h->rec_offset, h->rec_table[0], s));
      return collect_targets(h, nframes, det, ndet, corners, s, false);
    }
    HIPCHK(h, hipMemcpyAsync(h->d_stage, frames, (size_t)h->cfg.frame_bytes * nframes, hipMemcpyHostToDevice, s));
    d_frames = h->d_stage;
  }
  // Pipeline: the batch is cut into chunks that alternate over two streams, so that the latency-bound tail of
  // one chunk (target identification and pose: one wavefront per frame, a single dependency chain each) runs
  // under the bandwidth-bound head (ingest, threshold + corner pass) of the next.  Chunks only touch their own
  // frames' slices of the handle's buffers.  Per-stage times are recorded only on the single-pass path.
  int nchunks = h->pipeline_chunks;
  if (nchunks > 1) {
    const int min_chunk = 64;
    if (nframes / nchunks < min_chunk) nchunks = nframes / min_chunk;
  }
  if (nchunks > 1) {
    HIPCHK(h, hipEventRecord(h->pev[0], s));
    for (int k = 0; k < 2; ++k) HIPCHK(h, hipStreamWaitEvent(h->pstream[k], h->pev[0], 0));
    for (int c = 0; c < nchunks; ++c) {
      const int f0 = (int)((long long)nframes * c / nchunks), f1 = (int)((long long)nframes * (c + 1) / nchunks);
      rcc_handle v = handle_view(h, f0);
      v.want_thr = h->keep_bin ? 0 : 1;
      // the two-kernel form (variant 3) keeps its flat masks in ONE per-handle buffer indexed from frame 0: two chunks in
      // flight on the two streams would share its words, so a chunked batch runs the fused band kernel instead (bit-identical)
      if (v.dense_variant == 3) v.dense_variant = 1;
      hipStream_t cs = h->pstream[c & 1];
      hipError_t e = rcc_launch_ingest(&v, d_frames + (size_t)f0 * h->cfg.frame_bytes, f1 - f0, v.d_grey, cs);
      h->d_map = v.d_map; h->d_tilebox = v.d_tilebox; h->map_failed = v.map_failed; h->map_th = v.map_th;     // tables built by the first launch belong to the handle
      if (e == hipSuccess) e = rcc_launch_dense(&v, v.d_grey, f1 - f0, v.d_bin, v.d_cand, v.d_cand_count, cs);
      if (e != hipSuccess) { snprintf(h->err, sizeof(h->err), "%s", hipGetErrorString(e)); return RCC_ERR_DEVICE; }
      int r = launch_targets(&v, v.d_grey, v.d_bin, v.d_cand, v.d_cand_count, f1 - f0, cs, nullptr);
      if (r != RCC_OK) { snprintf(h->err, sizeof(h->err), "%s", v.err); return r; }
      h->bin_from_thr = v.bin_from_thr;
    }
    for (int k = 0; k < 2; ++k) {
      HIPCHK(h, hipEventRecord(h->pev[1 + k], h->pstream[k]));
      HIPCHK(h, hipStreamWaitEvent(s, h->pev[1 + k], 0));
    }
    for (float& m : h->last_ms) m = -1.0f;
    if (h->rec_table[0]) HIPCHK(h, rcc_launch_pack_records(h, nframes, h->rec_offset, h->rec_table[0], s));
    return collect_targets(h, nframes, det, ndet, corners, s, false);
  }
  HIPCHK(h, hipEventRecord(h->ev[0], s));
  HIPCHK(h, rcc_launch_ingest(h, d_frames, nframes, h->d_grey, s));
  HIPCHK(h, hipEventRecord(h->ev[1], s));
  // the stages after the dense pass sample the binary image at a few ring points per corner: the pass may leave
  // it as the compact per-tile threshold map (1/16 of the bytes); rcc_debug_fetch_images expands it on demand
  h->want_thr = h->keep_bin ? 0 : 1;
  HIPCHK(h, rcc_launch_dense(h, h->d_grey, nframes, h->d_bin, h->d_cand, h->d_cand_count, s));
  h->want_thr = 0;
  int r = launch_targets(h, h->d_grey, h->d_bin, h->d_cand, h->d_cand_count, nframes, s, &h->ev[2]);
  if (r != RCC_OK) return r;
  if (h->rec_table[0]) HIPCHK(h, rcc_launch_pack_records(h, nframes, h->rec_offset, h->rec_table[0], s));
  r = collect_targets(h, nframes, det, ndet, corners, s, true);
  if (r != RCC_OK) return r;
  (void)hipEventElapsedTime(&h->last_ms[0], h->ev[0], h->ev[1]);
  (void)hipEventElapsedTime(&h->last_ms[1], h->ev[1], h->ev[2]);
  return RCC_OK;
}

// ---- asynchronous form: submit batch k+1 before collecting batch k -----------------------------------------------
int rcc_detect_batch_submit(rcc_handle* h, const void* frames, int32_t nframes, int32_t frames_mem,
                            rcc_frame_corners* corners, void* stream)
{
  if (!h || !frames || nframes < 1) return RCC_ERR_ARG;
  if (frames_mem != RCC_MEM_HOST && frames_mem != RCC_MEM_DEVICE) return RCC_ERR_ARG;
  if (nframes > h->cfg.batch_capacity || !records_fit(h, nframes)) return RCC_ERR_CAPACITY;
  if (h->sub_head - h->sub_tail >= 2) return RCC_ERR_STATE;          // both result slots are in flight
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  // every submission shares the handle's one set of device buffers, so outstanding submissions must be ordered by ONE
  // stream: a second stream while a batch is in flight would let two batches race on them
  if (h->sub_head != h->sub_tail && h->sub_stream[(h->sub_head - 1) & 1u] != s) return RCC_ERR_STATE;
  if (detect_needs_bin(h)) { int rb = ensure_bin(h); if (rb != RCC_OK) return rb; }
  const int slot = (int)(h->sub_head & 1u);
  const int ring = (int)(h->sub_head % RCC_SUBT_RING);
  hipEvent_t* tev = h->sub_t_ev[ring];
  h->sub_t_seq[ring] = 0;
  const uint8_t* d_frames = (const uint8_t*)frames;
  bool piped = false;
  if (frames_mem == RCC_MEM_HOST) {
    { int rs = ensure_stage(h, (size_t)h->cfg.frame_bytes * nframes, h->sub_head == h->sub_tail); if (rs != RCC_OK) return rs; }
    const int per = host_chunk_frames(h, nframes);
    if (per > 0) {
      HIPCHK(h, hipEventRecord(tev[0], s));
      int r = launch_host_pipeline(h, (const uint8_t*)frames, nframes, per, s);
      if (r != RCC_OK) return r;
      piped = true;
    } else {
      HIPCHK(h, hipMemcpyAsync(h->d_stage, frames, (size_t)h->cfg.frame_bytes * nframes, hipMemcpyHostToDevice, s));
      d_frames = h->d_stage;
    }
  }
  const bool fid = h->cfg.target_kind == RCC_TARGET_FIDUCIAL;
  const int slots = h->cfg.max_targets;
  if (!piped) {
    HIPCHK(h, hipEventRecord(tev[0], s));
    HIPCHK(h, rcc_launch_ingest(h, d_frames, nframes, h->d_grey, s));
    HIPCHK(h, hipEventRecord(tev[1], s));
    h->want_thr = h->keep_bin ? 0 : 1;
    HIPCHK(h, rcc_launch_dense(h, h->d_grey, nframes, h->d_bin, h->d_cand, h->d_cand_count, s));
    h->want_thr = 0;
    { int rw = wait_corner_copy(h, s); if (rw != RCC_OK) return rw; }
#ifdef RCC_EXPERIMENTS
    // experiment (rcc_set_tail_overlap): the previous batch's lattice + pose kernel may still be running on the tail stream -- the
    // list stage is the first kernel that writes its buffers again
    if (h->tail_pending) { HIPCHK(h, hipStreamWaitEvent(s, h->tail_done[h->tail_pending - 1], 0)); h->tail_pending = 0; }
    if (h->tail_overlap && !fid && h->fuse_grid_pnp && rcc_grid_pnp_applicable(h) && !h->rec_table[slot] && !corners) {
      // S: list, sub-pixel, validation.  T (behind them): a marker, then lattice + pose, then the records to the host.  S goes on
      // behind the MARKER only, so the next batch's ingest pass -- queued on S by the next submission -- starts when the lattice +
      // pose kernel has just been handed to the device, and runs beside it.
      HIPCHK(h, hipEventRecord(tev[2], s));
      HIPCHK(h, rcc_launch_list(h, h->d_cand, h->d_cand_count, nframes, s));
      HIPCHK(h, rcc_launch_subpix(h, h->d_grey, nframes, s));
      HIPCHK(h, rcc_launch_validate(h, h->d_grey, h->d_bin, nframes, s));
      HIPCHK(h, hipEventRecord(tev[3], s));
      HIPCHK(h, hipEventRecord(h->tail_val[slot], s));
      hipStream_t t = h->tail_stream;
      HIPCHK(h, hipStreamWaitEvent(t, h->tail_val[slot], 0));
      HIPCHK(h, rcc_launch_marker(t));
      HIPCHK(h, hipEventRecord(h->tail_mark[slot], t));
      HIPCHK(h, rcc_launch_grid_pnp_only(h, nframes, t, 0));
      HIPCHK(h, hipEventRecord(tev[4], t));
      rcc_detection* hd = slot ? h->h_det2 : h->h_det;
      int32_t* hn = slot ? h->h_ndet2 : h->h_ndet;
      HIPCHK(h, hipMemcpyAsync(hd, h->d_det, sizeof(rcc_detection) * (size_t)nframes, hipMemcpyDeviceToHost, t));
      HIPCHK(h, hipMemcpyAsync(hn, h->d_ndet, sizeof(int32_t) * (size_t)nframes, hipMemcpyDeviceToHost, t));
      HIPCHK(h, hipEventRecord(tev[5], t));
      HIPCHK(h, hipEventRecord(h->sub_ev[slot], t));
      HIPCHK(h, hipEventRecord(h->tail_done[slot], t));
      h->tail_pending = slot + 1;
      HIPCHK(h, hipStreamWaitEvent(s, h->tail_mark[slot], 0));
      h->sub_has_fc[slot] = 0;
      h->sub_t_seq[ring] = h->sub_head + 1u;
      h->sub_t_staged[ring] = 1;
      h->sub_t_stream[ring] = s;
      h->sub_nframes[slot] = nframes;
      h->sub_stream[slot] = s;
      ++h->sub_head;
      for (float& m : h->last_ms) m = -1.0f;
      return RCC_OK;
    }
#endif
    int r = launch_targets(h, h->d_grey, h->d_bin, h->d_cand, h->d_cand_count, nframes, s, &tev[2]);
    if (r != RCC_OK) return r;
  }
  if (h->rec_table[slot]) HIPCHK(h, rcc_launch_pack_records(h, nframes, h->rec_offset, h->rec_table[slot], s));
  rcc_detection* hd = slot ? h->h_det2 : h->h_det;
  int32_t* hn = slot ? h->h_ndet2 : h->h_ndet;
  HIPCHK(h, hipMemcpyAsync(hd, h->d_det, sizeof(rcc_detection) * (size_t)nframes * (fid ? slots : 1), hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipMemcpyAsync(hn, h->d_ndet, sizeof(int32_t) * (size_t)nframes, hipMemcpyDeviceToHost, s));
  h->sub_has_fc[slot] = corners ? 1 : 0;
  if (corners) {
    // both landing areas with the first submission that asks for tables (a pinned allocation of 6 MB takes ~7 ms: once, not twice)
    for (int k = 0; k < 2; ++k)
      if (!h->h_fc[k] && hipHostMalloc((void**)&h->h_fc[k], sizeof(rcc_frame_corners) * (size_t)h->cfg.batch_capacity) != hipSuccess) {
        h->h_fc[k] = nullptr;
        return RCC_ERR_NOMEM;
      }
    h->sub_fc_dst[slot] = corners;
    HIPCHK(h, hipEventRecord(h->fc_ready[slot], s));
    HIPCHK(h, hipStreamWaitEvent(h->fc_stream, h->fc_ready[slot], 0));
    HIPCHK(h, hipMemcpyAsync(h->h_fc[slot], h->d_fc, sizeof(rcc_frame_corners) * (size_t)nframes, hipMemcpyDeviceToHost, h->fc_stream));
    HIPCHK(h, hipEventRecord(h->fc_done[slot], h->fc_stream));
    h->fc_pending = slot + 1;
  }
  HIPCHK(h, hipEventRecord(tev[5], s));
  HIPCHK(h, hipEventRecord(h->sub_ev[slot], s));
  h->sub_t_seq[ring] = h->sub_head + 1u;
  h->sub_t_staged[ring] = piped ? 0 : 1;
  h->sub_t_stream[ring] = s;
  h->sub_nframes[slot] = nframes;
  h->sub_stream[slot] = s;
  ++h->sub_head;
  for (float& m : h->last_ms) m = -1.0f;
  return RCC_OK;
}

int rcc_detect_batch_collect(rcc_handle* h, rcc_detection* det, int32_t* ndet)
{
  if (!h) return RCC_ERR_ARG;
  if (ndet) *ndet = 0;
  if (h->sub_head == h->sub_tail) return RCC_ERR_STATE;              // nothing submitted
  HIPCHK(h, hipSetDevice(h->device));
  const int slot = (int)(h->sub_tail & 1u);
  HIPCHK(h, hipEventSynchronize(h->sub_ev[slot]));
  if (h->sub_has_fc[slot]) {          // the corner tables came over on the copy stream, into the slot's pinned landing area
    HIPCHK(h, hipEventSynchronize(h->fc_done[slot]));
    if (h->fc_pending == slot + 1) h->fc_pending = 0;
    memcpy(h->sub_fc_dst[slot], h->h_fc[slot], sizeof(rcc_frame_corners) * (size_t)h->sub_nframes[slot]);
  }
  const bool fid = h->cfg.target_kind == RCC_TARGET_FIDUCIAL;
  const int slots = h->cfg.max_targets, nframes = h->sub_nframes[slot];
  const rcc_detection* hd = slot ? h->h_det2 : h->h_det;
  const int32_t* hn = slot ? h->h_ndet2 : h->h_ndet;
  int n = 0;
  for (int f = 0; f < nframes; ++f) {
    const int k = hn[f] < slots ? hn[f] : slots;
    for (int q = 0; q < k; ++q) {
      if (det) { det[n] = hd[(size_t)f * (fid ? slots : 1) + q]; det[n].frame = f; }
      ++n;
    }
  }
  if (ndet) *ndet = n;
  // what the streaming form records per batch: the device time from the moment the stream reached it to its records in pinned
  // memory, the device's idle time in front of it (> 0: the host submitted late), and the stages as they ran INSIDE this step
  {
    const unsigned sn = h->sub_tail;
    const int ring = (int)(sn % RCC_SUBT_RING), prev = (int)((sn + RCC_SUBT_RING - 1) % RCC_SUBT_RING);
    for (float& m : h->last_step_ms) m = -1.0f;
    if (h->sub_t_seq[ring] == sn + 1u) {
      hipEvent_t* tev = h->sub_t_ev[ring];
      if (hipEventElapsedTime(&h->last_step_ms[0], tev[0], tev[5]) != hipSuccess) h->last_step_ms[0] = -1.0f;
      if (sn > 0 && h->sub_t_seq[prev] == sn && h->sub_stream[slot] == h->sub_t_stream[prev]) {
        if (hipEventElapsedTime(&h->last_step_ms[1], h->sub_t_ev[prev][5], tev[0]) != hipSuccess) h->last_step_ms[1] = -1.0f;
      }
      if (h->sub_t_staged[ring])
        for (int i = 0; i < 5; ++i) {
          if (hipEventElapsedTime(&h->last_step_ms[2 + i], tev[i], tev[i + 1]) != hipSuccess) h->last_step_ms[2 + i] = -1.0f;
          h->last_ms[i] = h->last_step_ms[2 + i];
        }
    }
    (void)hipGetLastError();
  }
  h->sub_nframes[slot] = 0;
  ++h->sub_tail;
  return RCC_OK;
}

// record tables for the exchange between ranks (k_records.hip)
int rcc_set_record_tables(rcc_handle* h, double* d_table0, double* d_table1, int32_t capacity_slots, int32_t frame_offset)
{
  if (!h || (d_table1 && !d_table0) || (d_table0 && capacity_slots < 1)) return RCC_ERR_ARG;
  if (h->sub_head != h->sub_tail) return RCC_ERR_STATE;
  h->rec_table[0] = d_table0; h->rec_table[1] = d_table1 ? d_table1 : d_table0;
  h->rec_capacity = d_table0 ? capacity_slots : 0;
  h->rec_offset = frame_offset;
  return RCC_OK;
}
// a batch whose records do not fit the caller's tables is refused before anything is launched
static bool records_fit(const rcc_handle* h, int nframes)
{
  return !h->rec_table[0] || rcc_record_slots(h, nframes) <= h->rec_capacity;
}
int rcc_record_slots(const rcc_handle* h, int32_t nframes)
{
  if (!h || nframes < 0) return RCC_ERR_ARG;
  return nframes * (h->cfg.target_kind == RCC_TARGET_FIDUCIAL ? h->cfg.max_targets : 1);
}

int rcc_set_fuse_grid_pnp(rcc_handle* h, int on)
{
  if (!h) return RCC_ERR_ARG;
  int p = h->fuse_grid_pnp;
  h->fuse_grid_pnp = on ? 1 : 0;
  return p;
}

int rcc_set_keep_binary(rcc_handle* h, int on)
{
  if (!h) return RCC_ERR_ARG;
  int p = h->keep_bin;
  h->keep_bin = on ? 1 : 0;
  return p;
}

// host-resident batches: frames per chunk of the copy / compute pipeline (0 automatic, < 0 one copy then the kernels);
// returns the previous setting
int rcc_set_subpix_grid(rcc_handle* h, int width)
{
  if (!h || width < 0) return RCC_ERR_ARG;
  int p = h->subpix_grid;
  h->subpix_grid = width;
  return p;
}
int rcc_set_host_chunk(rcc_handle* h, int frames_per_chunk)
{
  if (!h) return RCC_ERR_ARG;
  const int p = h->host_chunk_frames;
  h->host_chunk_frames = frames_per_chunk;
  return p;
}

int rcc_set_pipeline(rcc_handle* h, int nchunks)
{
  if (!h || nchunks < 0 || nchunks > 64) return RCC_ERR_ARG;
  int p = h->pipeline_chunks;
  h->pipeline_chunks = nchunks;
  return p;
}

// debug/parity taps: copy the handle's intermediate lists of the last batch to the host
int rcc_debug_fetch_lists(rcc_handle* h, int32_t nframes, void* pre /*B*256 cand*/, int32_t* npre,
                          double* pre_xy, void* kept, double* kept_xy)
{
  if (!h || nframes < 0 || nframes > h->cfg.batch_capacity) return RCC_ERR_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  const size_t n = (size_t)nframes;
  const size_t kc = (size_t)h->kept_cap;
  if (pre) HIPCHK(h, hipMemcpy(pre, h->d_pre, n * kc * sizeof(rcc_cand), hipMemcpyDeviceToHost));
  if (npre) HIPCHK(h, hipMemcpy(npre, h->d_npre, n * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (pre_xy) HIPCHK(h, hipMemcpy(pre_xy, h->d_pre_xy, n * kc * 2 * sizeof(double), hipMemcpyDeviceToHost));
  if (kept && h->cfg.target_kind == RCC_TARGET_CHECKERBOARD) HIPCHK(h, hipMemcpy(kept, h->d_kept, n * RCC_MAX_KEPT * sizeof(rcc_cand), hipMemcpyDeviceToHost));
  if (kept_xy && h->cfg.target_kind == RCC_TARGET_CHECKERBOARD) HIPCHK(h, hipMemcpy(kept_xy, h->d_kept_xy, n * RCC_MAX_KEPT * 2 * sizeof(double), hipMemcpyDeviceToHost));
  return RCC_OK;
}
int rcc_debug_fetch_images(rcc_handle* h, int32_t nframes, void* grey, void* bin, void* cand, int32_t* cand_count)
{
  if (!h || nframes < 0 || nframes > h->cfg.batch_capacity) return RCC_ERR_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  const size_t n = (size_t)nframes, px = (size_t)h->cfg.width * h->cfg.height;
  if (grey) HIPCHK(h, hipMemcpy(grey, h->d_grey, n * px, hipMemcpyDeviceToHost));
  if (bin) {
    { int rb = ensure_bin(h); if (rb != RCC_OK) return rb; }
    if (h->bin_from_thr) {
      HIPCHK(h, rcc_launch_expand_bin(h, h->d_grey, nframes, h->d_bin, h->stream));
      HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    HIPCHK(h, hipMemcpy(bin, h->d_bin, n * px, hipMemcpyDeviceToHost));
  }
  if (cand) HIPCHK(h, hipMemcpy(cand, h->d_cand, n * (size_t)h->cfg.max_candidates * sizeof(rcc_cand), hipMemcpyDeviceToHost));
  if (cand_count) HIPCHK(h, hipMemcpy(cand_count, h->d_cand_count, n * sizeof(int32_t), hipMemcpyDeviceToHost));
  return RCC_OK;
}

// profiling aid: dword-per-lane streaming copy of nbytes (device pointers), see k_dense_fast.hip
int rcc_debug_calib_copy(rcc_handle* h, const void* d_src, void* d_dst, int64_t nbytes)
{
  if (!h || !d_src || !d_dst || nbytes < 4) return RCC_ERR_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, rcc_launch_calib_copy(d_src, d_dst, (size_t)nbytes, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return RCC_OK;
}

// measurement aid: mean time of a plain 16 B/lane streaming copy of nbytes (device pointers), the yardstick the
// bandwidth-bound passes are compared with on the same box (bench.py's roofline leg)
int rcc_time_copy(rcc_handle* h, const void* d_src, void* d_dst, int64_t nbytes, int32_t reps, float* mean_ms)
{
  if (!h || !d_src || !d_dst || nbytes < 16 || (nbytes & 15) || reps < 1 || !mean_ms) return RCC_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(d_src) | reinterpret_cast<uintptr_t>(d_dst)) & 15) return RCC_ERR_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t s = h->stream;
  HIPCHK(h, rcc_launch_copy_x4(d_src, d_dst, (size_t)nbytes, s));
  HIPCHK(h, hipEventRecord(h->ev[6], s));
  for (int r = 0; r < reps; ++r) HIPCHK(h, rcc_launch_copy_x4(d_src, d_dst, (size_t)nbytes, s));
  HIPCHK(h, hipEventRecord(h->ev[7], s));
  HIPCHK(h, hipEventSynchronize(h->ev[7]));
  float ms = 0.0f;
  HIPCHK(h, hipEventElapsedTime(&ms, h->ev[6], h->ev[7]));
  *mean_ms = ms / (float)reps;
  return RCC_OK;
}

// measurement aid (k_probe.hip): the engine clock the chip holds under a vector-issue load and what a wave-instruction of the
// threshold + corner pass's instruction classes costs per SIMD, measured on this device by this call
int rcc_debug_measure_clock(rcc_handle* h, int32_t waves_per_simd, float ms_target, double* out6)
{
  if (!h || !out6 || waves_per_simd < 1 || waves_per_simd > 8 || !(ms_target > 0.0f) || ms_target > 1000.0f) return RCC_ERR_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  hipDeviceProp_t pr;
  HIPCHK(h, hipGetDeviceProperties(&pr, h->device));
  const int blocks = pr.multiProcessorCount * waves_per_simd;       // 256 threads = four waves = one per SIMD of a CU
  const size_t nw = (size_t)blocks * 4;
  unsigned long long* d_st = nullptr;
  unsigned* d_sink = nullptr;
  if (hipMalloc((void**)&d_st, nw * 2 * sizeof(unsigned long long)) != hipSuccess) return RCC_ERR_NOMEM;
  if (hipMalloc((void**)&d_sink, 64) != hipSuccess) { (void)hipFree(d_st); return RCC_ERR_NOMEM; }
  hipStream_t s = h->stream;
  int rc = RCC_OK;
  float ms = 0.0f;
  int iters = 256;
  std::vector<unsigned long long> st(nw * 2);
  auto run = [&](int it) -> bool {
    return hipEventRecord(h->ev[6], s) == hipSuccess && rcc_launch_issue_probe(blocks, it, d_st, d_sink, s) == hipSuccess &&
           hipEventRecord(h->ev[7], s) == hipSuccess && hipEventSynchronize(h->ev[7]) == hipSuccess &&
           hipEventElapsedTime(&ms, h->ev[6], h->ev[7]) == hipSuccess;
  };
  if (!run(iters) || !run(iters)) rc = RCC_ERR_DEVICE;                 // the second launch sizes the measured one
  if (rc == RCC_OK) {
    double want = (double)iters * (double)ms_target / (ms > 1e-3f ? ms : 1e-3f);
    if (want < 64.0) want = 64.0;
    if (want > 4.0e6) want = 4.0e6;
    iters = (int)want;
    if (!run(iters) || hipMemcpy(st.data(), d_st, nw * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) rc = RCC_ERR_DEVICE;
  }
  (void)hipFree(d_st);
  (void)hipFree(d_sink);
  if (rc != RCC_OK) { snprintf(h->err, sizeof(h->err), "issue probe: %s", hipGetErrorString(hipGetLastError())); return rc; }
  std::vector<double> mhz;
  mhz.reserve(nw);
  for (size_t w = 0; w < nw; ++w)
    if (st[2 * w + 1] > 0) mhz.push_back(100.0 * (double)st[2 * w] / (double)st[2 * w + 1]);   // s_memrealtime ticks at 100 MHz
  if (mhz.empty()) return RCC_ERR_DEVICE;
  std::sort(mhz.begin(), mhz.end());
  const double ns = (double)ms * 1e6 / ((double)iters * RCC_PROBE_INSTS_PER_ITER * (double)waves_per_simd);
  out6[0] = mhz[mhz.size() / 2];
  out6[1] = ns;
  out6[2] = ns * 1e-3 * out6[0];
  out6[3] = (double)ms;
  out6[4] = mhz.front();
  out6[5] = mhz.back();
  return RCC_OK;
}

#ifdef RCC_EXPERIMENTS
#ifdef RCC_EXPERIMENTS
int rcc_set_tail_overlap(rcc_handle* h, int mode)
{
  if (!h || mode < 0 || mode > 1) return RCC_ERR_ARG;
  if (h->sub_head != h->sub_tail) return RCC_ERR_STATE;
  HIPCHK(h, hipSetDevice(h->device));
  if (mode && !h->tail_stream) {
    HIPCHK(h, hipStreamCreateWithFlags(&h->tail_stream, hipStreamNonBlocking));
    for (int k = 0; k < 2; ++k) {
      HIPCHK(h, hipEventCreateWithFlags(&h->tail_val[k], hipEventDisableTiming));
      HIPCHK(h, hipEventCreateWithFlags(&h->tail_mark[k], hipEventDisableTiming));
      HIPCHK(h, hipEventCreateWithFlags(&h->tail_done[k], hipEventDisableTiming));
    }
  }
  h->tail_overlap = mode;
  return RCC_OK;
}
#endif
int rcc_set_dense_fmod(rcc_handle* h, int m)
{
  if (!h || m < 0) return RCC_ERR_ARG;
  h->dense_fmod = m;
  return RCC_OK;
}
// experiment: per-frame phase time stamps of k_grid_pnp (8 x int64 per frame, wall_clock64 ticks of 10 ns): NULL switches off
hipError_t rcc_set_grid_trace(long long* d_buf);
int rcc_debug_grid_trace(rcc_handle* h, void* d_buf)
{
  if (!h) return RCC_ERR_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, rcc_set_grid_trace((long long*)d_buf));
  return RCC_OK;
}
// experiment: do the ingest pass (bandwidth-bound) and the threshold+corner pass (issue-bound) overlap when they are
// launched on two streams over independent buffers?  mode 0: back to back on one stream; 1: concurrently on two streams;
// 2: as two roles of one launch (k_mix.hip); 3 / 4: the ingest / the threshold+corner pass alone.
int rcc_debug_overlap(rcc_handle* h, const void* d_frames, int32_t nframes, void* d_grey_out, const void* d_grey_in,
                      void* d_cand, void* d_cand_count, int32_t mode, int32_t reps, float* mean_ms)
{
  if (!h || !d_frames || !d_grey_out || !d_grey_in || !d_cand || !d_cand_count || nframes < 1 || reps < 1 || !mean_ms) return RCC_ERR_ARG;
  if (nframes > h->cfg.batch_capacity) return RCC_ERR_CAPACITY;
  HIPCHK(h, hipSetDevice(h->device));
  if (detect_needs_bin(h)) { int rb = ensure_bin(h); if (rb != RCC_OK) return rb; }
  hipStream_t s = h->stream, a = h->pstream[0], b = h->pstream[1];
  HIPCHK(h, hipStreamSynchronize(s));
  HIPCHK(h, hipEventRecord(h->ev[6], s));
  for (int r = 0; r < reps; ++r) {
    HIPCHK(h, hipMemsetAsync(d_cand_count, 0, sizeof(int32_t) * (size_t)nframes, s));
    if (mode == 0 || mode == 3 || mode == 4) {
      // 0: back to back; 3: the ingest pass alone; 4: the threshold + corner pass alone
      if (mode != 4) HIPCHK(h, rcc_launch_ingest(h, (const uint8_t*)d_frames, nframes, (uint8_t*)d_grey_out, s));
      h->want_thr = 1;
      if (mode != 3) HIPCHK(h, rcc_launch_dense(h, (const uint8_t*)d_grey_in, nframes, h->d_bin, (rcc_cand*)d_cand, (int32_t*)d_cand_count, s));
    } else if (mode == 2) {
      // both passes as roles of one launch (k_mix.hip)
      bool done = false;
      HIPCHK(h, rcc_launch_mix(h, (const uint8_t*)d_frames, nframes, (uint8_t*)d_grey_out, (const uint8_t*)d_grey_in, nframes, h->d_thr,
                               (rcc_cand*)d_cand, (int32_t*)d_cand_count, s, &done));
      if (!done) return RCC_ERR_ARG;
    } else {
      HIPCHK(h, hipEventRecord(h->pev[0], s));
      HIPCHK(h, hipStreamWaitEvent(a, h->pev[0], 0));
      HIPCHK(h, hipStreamWaitEvent(b, h->pev[0], 0));
      HIPCHK(h, rcc_launch_ingest(h, (const uint8_t*)d_frames, nframes, (uint8_t*)d_grey_out, a));
      h->want_thr = 1;
      HIPCHK(h, rcc_launch_dense(h, (const uint8_t*)d_grey_in, nframes, h->d_bin, (rcc_cand*)d_cand, (int32_t*)d_cand_count, b));
      HIPCHK(h, hipEventRecord(h->pev[1], a));
      HIPCHK(h, hipEventRecord(h->pev[2], b));
      HIPCHK(h, hipStreamWaitEvent(s, h->pev[1], 0));
      HIPCHK(h, hipStreamWaitEvent(s, h->pev[2], 0));
    }
  }
  h->want_thr = 0;
  HIPCHK(h, hipEventRecord(h->ev[7], s));
  HIPCHK(h, hipEventSynchronize(h->ev[7]));
  float ms = 0.0f;
  HIPCHK(h, hipEventElapsedTime(&ms, h->ev[6], h->ev[7]));
  *mean_ms = ms / (float)reps;
  return RCC_OK;
}

#endif  // RCC_EXPERIMENTS

// ---- solvePnP / Rodrigues drop-ins -----------------------------------------------------------------
static int ensure_pnp_buf(rcc_handle* h, size_t bytes)
{
  if (bytes <= h->pnp_buf_bytes) return RCC_OK;
  if (h->d_pnp_buf) (void)hipFree(h->d_pnp_buf);
  h->d_pnp_buf = nullptr;
  h->pnp_buf_bytes = 0;
  size_t cap = bytes + bytes / 2 + 4096;
  if (hipMalloc((void**)&h->d_pnp_buf, cap) != hipSuccess) return RCC_ERR_NOMEM;
  h->pnp_buf_bytes = cap;
  return RCC_OK;
}

int rcc_solve_pnp_batch(rcc_handle* h, const double* obj, const double* img, const int32_t* npts,
                        int32_t ntargets, const double* K, const double* D, int32_t dist_model,
                        double* rvec, double* tvec, double* rms, int32_t* status, int32_t* iters)
{
  if (!h || ntargets < 0 || (ntargets > 0 && (!obj || !img || !npts || !rvec || !tvec))) return RCC_ERR_ARG;
  if (ntargets == 0) return RCC_OK;
  if (dist_model != RCC_DIST_NONE && dist_model != RCC_DIST_PLUMB_BOB) return RCC_ERR_UNSUPPORTED;
  HIPCHK(h, hipSetDevice(h->device));
  std::vector<int32_t> off((size_t)ntargets);
  size_t total = 0;
  int maxpts = 0;
  for (int t = 0; t < ntargets; ++t) {
    if (npts[t] < 0) return RCC_ERR_ARG;
    off[t] = (int32_t)total;
    total += (size_t)npts[t];
    if (npts[t] > maxpts) maxpts = npts[t];
  }
  h->pnp_wave_hint = (maxpts > 8 && h->pnp_variant != 0) ? 1 : 0;
  const size_t T = (size_t)ntargets;
  // layout (8-byte units): obj[3*total] img[2*total] rvec[3T] tvec[3T] rms[T] | int32: off[T] npts[T] status[T] iters[T]
  const size_t nd = 5 * total + 7 * T;
  const size_t bytes = nd * sizeof(double) + 4 * T * sizeof(int32_t);
  int r = ensure_pnp_buf(h, bytes);
  if (r != RCC_OK) return r;
  double* d_obj = h->d_pnp_buf;
  double* d_img = d_obj + 3 * total;
  double* d_rvec = d_img + 2 * total;
  double* d_tvec = d_rvec + 3 * T;
  double* d_rms = d_tvec + 3 * T;
  int32_t* d_off = (int32_t*)(d_rms + T);
  int32_t* d_npts = d_off + T;
  int32_t* d_status = d_npts + T;
  int32_t* d_iters = d_status + T;
  hipStream_t s = h->stream;
  HIPCHK(h, hipMemcpyAsync(d_obj, obj, 3 * total * sizeof(double), hipMemcpyHostToDevice, s));
  HIPCHK(h, hipMemcpyAsync(d_img, img, 2 * total * sizeof(double), hipMemcpyHostToDevice, s));
  HIPCHK(h, hipMemcpyAsync(d_off, off.data(), T * sizeof(int32_t), hipMemcpyHostToDevice, s));
  HIPCHK(h, hipMemcpyAsync(d_npts, npts, T * sizeof(int32_t), hipMemcpyHostToDevice, s));
  rcc_cam cam;
  const double* Kp = K ? K : h->cfg.K;
  const double* Dp = D ? D : h->cfg.D;
  cam.fx = Kp[0]; cam.cx = Kp[2]; cam.fy = Kp[4]; cam.cy = Kp[5];
  for (int i = 0; i < 8; ++i) cam.D[i] = (i < 5) ? Dp[i] : 0.0;
  cam.model = dist_model;
  cam.solver = h->pnp_solver;
  HIPCHK(h, rcc_launch_pnp_generic(h, d_obj, d_img, d_off, d_npts, ntargets, cam, d_rvec, d_tvec, d_rms, d_status, d_iters, s));
  HIPCHK(h, hipMemcpyAsync(rvec, d_rvec, 3 * T * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipMemcpyAsync(tvec, d_tvec, 3 * T * sizeof(double), hipMemcpyDeviceToHost, s));
  if (rms) HIPCHK(h, hipMemcpyAsync(rms, d_rms, T * sizeof(double), hipMemcpyDeviceToHost, s));
  if (status) HIPCHK(h, hipMemcpyAsync(status, d_status, T * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  if (iters) HIPCHK(h, hipMemcpyAsync(iters, d_iters, T * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  return RCC_OK;
}

// test tap (host pointers): out[59] = H[9], initial pose[6], JtJ[36], Jte[6], |e|^2, status*10+ok
int rcc_debug_pnp_probe(rcc_handle* h, const double* obj, const double* img, int32_t n, const double* K,
                        const double* D, int32_t dist_model, double* out)
{
  if (!h || !obj || !img || n < 4 || !K || !D || !out) return RCC_ERR_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  int r = ensure_pnp_buf(h, (size_t)(5 * n + 64) * sizeof(double));
  if (r != RCC_OK) return r;
  double* d_obj = h->d_pnp_buf;
  double* d_img = d_obj + 3 * n;
  double* d_out = d_img + 2 * n;
  hipStream_t s = h->stream;
  HIPCHK(h, hipMemcpyAsync(d_obj, obj, 3 * (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
  HIPCHK(h, hipMemcpyAsync(d_img, img, 2 * (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
  rcc_cam cam;
  cam.fx = K[0]; cam.cx = K[2]; cam.fy = K[4]; cam.cy = K[5];
  for (int i = 0; i < 8; ++i) cam.D[i] = (i < 5) ? D[i] : 0.0;
  cam.model = dist_model;
  cam.solver = h->pnp_solver;
  HIPCHK(h, rcc_launch_pnp_probe(d_obj, d_img, n, cam, d_out, s));
  HIPCHK(h, hipMemcpyAsync(out, d_out, 59 * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  return RCC_OK;
}

static int rodrigues_batch(rcc_handle* h, int dir, const double* in, int32_t n, double* out)
{
  if (!h || n < 0 || (n > 0 && (!in || !out))) return RCC_ERR_ARG;
  if (n == 0) return RCC_OK;
  HIPCHK(h, hipSetDevice(h->device));
  const size_t nin = (dir == 0 ? 3 : 9) * (size_t)n, nout = (dir == 0 ? 9 : 3) * (size_t)n;
  int r = ensure_pnp_buf(h, (nin + nout) * sizeof(double));
  if (r != RCC_OK) return r;
  hipStream_t s = h->stream;
  HIPCHK(h, hipMemcpyAsync(h->d_pnp_buf, in, nin * sizeof(double), hipMemcpyHostToDevice, s));
  HIPCHK(h, rcc_launch_rodrigues(dir, h->d_pnp_buf, n, h->d_pnp_buf + nin, s));
  HIPCHK(h, hipMemcpyAsync(out, h->d_pnp_buf + nin, nout * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  return RCC_OK;
}
int rcc_rodrigues_v2m_batch(rcc_handle* h, const double* rvec, int32_t n, double* R9) { return rodrigues_batch(h, 0, rvec, n, R9); }
int rcc_rodrigues_m2v_batch(rcc_handle* h, const double* R9, int32_t n, double* rvec) { return rodrigues_batch(h, 1, R9, n, rvec); }

// ---- synthetic camera ------------------------------------------------------------------------------
int rcc_synth_render_batch(rcc_handle* h, const rcc_synth_params* sp, const double* poses,
                           int32_t nframes, int32_t first_frame_index, void* d_frames, void* stream)
{
  if (!h || !sp || sp->struct_size != sizeof(rcc_synth_params) || !poses || !d_frames || nframes < 0) return RCC_ERR_ARG;
  if (nframes == 0) return RCC_OK;
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  std::vector<double> hinv(9 * (size_t)nframes);
  for (int f = 0; f < nframes; ++f) {
    const double* p = poses + 6 * (size_t)f;
    double R[9];
    rccpnp::rodrigues_v2m(p, R, nullptr);
    const double M[9] = { R[0], R[1], p[3], R[3], R[4], p[4], R[6], R[7], p[5] };
    const double d = rccpnp::mat3_det(M);
    const double q = 1.0 / d;
    double* I = &hinv[9 * (size_t)f];
    I[0] = (M[4] * M[8] - M[5] * M[7]) * q; I[1] = (M[2] * M[7] - M[1] * M[8]) * q; I[2] = (M[1] * M[5] - M[2] * M[4]) * q;
    I[3] = (M[5] * M[6] - M[3] * M[8]) * q; I[4] = (M[0] * M[8] - M[2] * M[6]) * q; I[5] = (M[2] * M[3] - M[0] * M[5]) * q;
    I[6] = (M[3] * M[7] - M[4] * M[6]) * q; I[7] = (M[1] * M[6] - M[0] * M[7]) * q; I[8] = (M[0] * M[4] - M[1] * M[3]) * q;
  }
  int r = ensure_pnp_buf(h, hinv.size() * sizeof(double));
  if (r != RCC_OK) return r;
  HIPCHK(h, hipMemcpyAsync(h->d_pnp_buf, hinv.data(), hinv.size() * sizeof(double), hipMemcpyHostToDevice, s));
  {
    hipError_t e = rcc_launch_synth(h, sp, h->d_pnp_buf, nframes, first_frame_index, (uint8_t*)d_frames, s);
    if (e == hipErrorInvalidValue) return RCC_ERR_ARG;          // malformed optics parameters (include/rcc.h)
    if (e == hipErrorOutOfMemory) return RCC_ERR_NOMEM;
    HIPCHK(h, e);
  }
  HIPCHK(h, hipStreamSynchronize(s));
  return RCC_OK;
}

}  // extern "C"
