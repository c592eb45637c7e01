// k_pnp.hip -- stage a7/a8 kernels: per-target pose (solvePnP ITERATIVE form) and Rodrigues.
//
// Drop-in target: cv::solvePnP(...) at real_preprocessing/src/camera_pose.cpp:163 and
// cv::Rodrigues at :164 (and :93,:116; opt_visualization.cpp:36).  Arithmetic in pnp_core.h.
//
// Two mappings, chosen by point count:
//   * few points (a 4-corner fiducial, camera_pose.cpp:152-161): one LANE per target.  The solve is
//     a chain of dependent fp64 operations with small factorizations in the middle and nothing to
//     spread over 4 points, so targets are the parallel axis (64 per wavefront).
//   * many points (the 48-corner board): one WAVEFRONT per target.  Lanes take the points; every
//     per-point accumulation (DLT 9x9, homography refinement 8x8, pose JtJ 6x6 / Jte) is a lane-local
//     partial sum + xor-butterfly, after which all lanes hold the same normal equations and run the
//     small dense algebra redundantly (wave-uniform, no divergence, no LDS).
// The result record layout is the C ABI's rcc_detection (what corner_detections.cpp:46-56 reads,
// plus the pose).
#include "rcc_internal.h"
// The solver routines of pnp_core.h are inlined (no RCC_PNP_NOINLINE): round 1 kept them out of line after a suspected
// hipcc -O3 miscompile that round 2 could not reproduce -- the fully inlined -O3 build passes every pose parity test
// (the one recorded failure was the test's own: the Rodrigues round trip is not unique beyond |r| = pi) and is faster
// (24 456 tag poses 0.45 -> 0.31 ms, board pose 0.27 -> 0.25 ms; profiles/r02_e_pnp_inline.txt).
#ifdef RCC_EXPERIMENTS
// experiment builds only: per-frame time stamps of k_grid_pnp's phases (rcc_debug_grid_trace): 24 slots per frame
__device__ long long* g_grid_trace = nullptr;
#define GTRACE(slot) do { if (g_grid_trace && lane == 0) g_grid_trace[(size_t)f * 24 + (slot)] = (long long)wall_clock64(); } while (0)
#define GTRACE_VAL(slot, v) do { if (g_grid_trace && lane == 0) g_grid_trace[(size_t)f * 24 + (slot)] = (long long)(v); } while (0)
// solver phases (pnp_core.h): k < 100 a time stamp in slot k; k >= 100: the refinement's iteration count into slot 13
#define RCC_PNP_PHASE(k) do { if (g_grid_trace && threadIdx.x == 0) { if ((k) >= 100) g_grid_trace[(size_t)blockIdx.x * 24 + 13] = (k) - 100; else g_grid_trace[(size_t)blockIdx.x * 24 + (k)] = (long long)wall_clock64(); } } while (0)
// inner steps: category accumulators (10-ns ticks) in the free tail of the wavefront's LDS workspace (n <= 48 points)
#define RCC_PNP_TIC() long long tic_ = (long long)wall_clock64()
#define RCC_PNP_TOC(cat) do { const long long now_ = (long long)wall_clock64(); if (g_grid_trace && par.first() == 0) ((long long*)(par.ws() + 308))[cat] += now_ - tic_; tic_ = (long long)wall_clock64(); } while (0)
#endif
#include "pnp_core.h"
#ifdef RCC_EXPERIMENTS
extern "C" hipError_t rcc_set_grid_trace(long long* d_buf) { return hipMemcpyToSymbol(HIP_SYMBOL(g_grid_trace), &d_buf, sizeof(d_buf)); }
#endif
#include "grid_frame.h"

static __device__ __forceinline__ rccpnp::Cam to_cam(const rcc_cam& c)
{
  rccpnp::Cam k;
  k.fx = c.fx; k.fy = c.fy; k.cx = c.cx; k.cy = c.cy;
  for (int i = 0; i < 5; ++i) k.k[i] = c.D[i];
  k.solver = c.solver;
  return k;
}

__global__ __launch_bounds__(64) void k_pnp_generic(const double* __restrict__ obj, const double* __restrict__ img,
                                                    const int32_t* __restrict__ off, const int32_t* __restrict__ npts,
                                                    int ntargets, rcc_cam cam, double* __restrict__ rvec,
                                                    double* __restrict__ tvec, double* __restrict__ rms,
                                                    int32_t* __restrict__ status, int32_t* __restrict__ iters)
{
  const int t = blockIdx.x * 64 + threadIdx.x;
  if (t >= ntargets) return;
  rccpnp::Pts p{ obj + 3 * (size_t)off[t], img + 2 * (size_t)off[t], npts[t] };
  double r[3], tv[3], e = 0.0;
  double wsl[rccpnp::PNP_WS];
  int it = 0;
  int st = rccpnp::solve_pnp(rccpnp::SerialPar{ wsl }, p, to_cam(cam), cam.model, r, tv, &e, &it);
  for (int k = 0; k < 3; ++k) { rvec[3 * t + k] = r[k]; tvec[3 * t + k] = tv[k]; }
  if (rms) rms[t] = e;
  if (status) status[t] = st;
  if (iters) iters[t] = it;
}

// one lane per frame: board corners (rcc_frame_corners) -> rcc_detection
__global__ __launch_bounds__(64) void k_pnp_board(const rcc_frame_corners* __restrict__ fc, int nframes,
                                                  const double* __restrict__ board_obj, int cols, int rows,
                                                  double square, int board_id, int reference_mode,
                                                  rcc_cam cam, double* __restrict__ img_scratch,
                                                  rcc_detection* __restrict__ det, int32_t* __restrict__ ndet)
{
  const int f = blockIdx.x * 64 + threadIdx.x;
  if (f >= nframes) return;
  const rcc_frame_corners* c = fc + f;
  const int need = cols * rows;
  if (c->status != 0 || c->ncorners != need) { ndet[f] = 0; return; }
  double* img = img_scratch + (size_t)f * 2 * RCC_MAX_BOARD_CORNERS;
  for (int k = 0; k < need; ++k) {
    double x = c->xy[k][0], y = c->xy[k][1];
    if (reference_mode) { x = (double)(int)x; y = (double)(int)y; }   // corner_detections.cpp:53-54
    img[2 * k] = x;
    img[2 * k + 1] = y;
  }
  rccpnp::Pts p{ board_obj, img, need };
  rcc_detection d;
  d.frame = f;
  d.id = board_id;
  d.hamming = 0;
  d.ncorners = need;
  d.size = square;
  const int idx[4] = { (rows - 1) * cols, (rows - 1) * cols + cols - 1, cols - 1, 0 };   // bl, br, tr, tl
  for (int k = 0; k < 4; ++k) { d.corners[k][0] = c->xy[idx[k]][0]; d.corners[k][1] = c->xy[idx[k]][1]; }
  int it = 0;
  double wsl[rccpnp::PNP_WS];
  d.pnp_status = rccpnp::solve_pnp(rccpnp::SerialPar{ wsl }, p, to_cam(cam), cam.model, d.rvec, d.tvec, &d.rms, &it);
  d.pnp_iters = it;
  det[f] = d;
  ndet[f] = 1;
}

// staging area of the matrix-core accumulation of the normal equations (WavePar::gram in pnp_core.h): an experiment that
// lost to the vector path, compiled in only with -DRCC_PNP_GRAM
#ifdef RCC_PNP_GRAM
#define RCC_GRAM_STAGE(name) __shared__ double name[rccpnp::WavePar::GRAM_ROWS * rccpnp::WavePar::GRAM_STRIDE]
#else
#define RCC_GRAM_STAGE(name) double* const name = nullptr
#endif
__device__ __forceinline__ unsigned gram_lds(const double* p) { return p ? (unsigned)(uintptr_t)p : rccpnp::WavePar::GRAM_NONE; }

// one wavefront per frame: pose of the board from its ordered corners.  XY(k, x, y) hands out corner k (row-major
// lattice index); img: 2 * need doubles the solver reads its image points from (per frame, any memory)
template <class XY>
__device__ __forceinline__ void board_pose_frame(const int f, const int lane, XY xy_of, double* __restrict__ img,
                                                 const double* __restrict__ board_obj, int cols, int rows,
                                                 double square, int board_id, int reference_mode, const rcc_cam& cam,
                                                 double* ws, double* gram_stage, rcc_detection* __restrict__ det, int32_t* __restrict__ ndet)
{
  const int need = cols * rows;
  for (int k = lane; k < need; k += 64) {
    double x, y;
    xy_of(k, x, y);
    if (reference_mode) { x = (double)(int)x; y = (double)(int)y; }   // corner_detections.cpp:53-54
    img[2 * k] = x;
    img[2 * k + 1] = y;
  }
  __syncthreads();
  rccpnp::Pts p{ board_obj, img, need };
  rccpnp::WavePar par{ lane, (unsigned)(uintptr_t)ws, gram_lds(gram_stage) };
  double r[3], tv[3], e = 0.0;
  int it = 0;
  const int st = rccpnp::solve_pnp(par, p, to_cam(cam), cam.model, r, tv, &e, &it);
  if (lane == 0) {
    rcc_detection d;
    d.frame = f;
    d.id = board_id;
    d.hamming = 0;
    d.ncorners = need;
    d.size = square;
    const int idx[4] = { (rows - 1) * cols, (rows - 1) * cols + cols - 1, cols - 1, 0 };   // bl, br, tr, tl
    for (int k = 0; k < 4; ++k) xy_of(idx[k], d.corners[k][0], d.corners[k][1]);
    for (int k = 0; k < 3; ++k) { d.rvec[k] = r[k]; d.tvec[k] = tv[k]; }
    d.rms = e;
    d.pnp_status = st;
    d.pnp_iters = it;
    det[f] = d;
    ndet[f] = 1;
  }
}

__global__ __launch_bounds__(64) void k_pnp_board_wave(const rcc_frame_corners* __restrict__ fc, int nframes,
                                                       const double* __restrict__ board_obj, int cols, int rows,
                                                       double square, int board_id, int reference_mode,
                                                       rcc_cam cam, double* __restrict__ img_scratch,
                                                       rcc_detection* __restrict__ det, int32_t* __restrict__ ndet)
{
  const int f = blockIdx.x;
  const int lane = threadIdx.x;
  const rcc_frame_corners* c = fc + f;
  if (c->status != 0 || c->ncorners != cols * rows) { if (lane == 0) ndet[f] = 0; return; }
  __shared__ double ws[rccpnp::PNP_WS];          // the wave-uniform matrices: one copy per wavefront, in LDS
  RCC_GRAM_STAGE(gst);
  board_pose_frame(f, lane, [&](int k, double& x, double& y) { x = c->xy[k][0]; y = c->xy[k][1]; },
                   img_scratch + (size_t)f * 2 * RCC_MAX_BOARD_CORNERS, board_obj, cols, rows, square, board_id, reference_mode, cam, ws, gst, det, ndet);
}

// a6 + a7 of one frame in one wavefront: lattice indexing of the validated corners (grid_frame.h), then the pose from the
// lattice it leaves in LDS.  Both stages are single dependency chains per frame; run as two kernels the second waits for the
// slowest frame of the first.  (a4.3, the validation, is a launch of its own: k_validate, one candidate per thread.)
__device__ __forceinline__ void grid_pnp_frame(int w, int h, const rcc_cand* __restrict__ kept, const double* __restrict__ kept_xy,
                                               int cols, int rows, rcc_frame_corners* __restrict__ fc,
                                               const double* __restrict__ board_obj, double square, int board_id, int reference_mode,
                                               const rcc_cam& cam, rcc_detection* __restrict__ det, int32_t* __restrict__ ndet)
{
  __shared__ grid_smem sm;
  __shared__ double ws[rccpnp::PNP_WS];
  __shared__ double s_img[2 * RCC_MAX_BOARD_CORNERS];
  RCC_GRAM_STAGE(gst);
  const int f = blockIdx.x;
  const int lane = threadIdx.x;
#ifdef RCC_EXPERIMENTS
  if (threadIdx.x < 8) ((long long*)(ws + 308))[threadIdx.x] = 0;
#endif
  GTRACE(0);
  const bool found = index_frame(sm, f, lane, w, h, kept, kept_xy, RCC_TARGET_CHECKERBOARD, cols, rows, fc);
  __syncthreads();
  GTRACE(5);
  if (!found) { if (lane == 0) ndet[f] = 0; return; }       // wave-uniform
  board_pose_frame(f, lane, [&](int k, double& x, double& y) { const int o = sm.order[k]; x = sm.xy[2 * o]; y = sm.xy[2 * o + 1]; },
                   s_img, board_obj, cols, rows, square, board_id, reference_mode, cam, ws, gst, det, ndet);
  GTRACE(6);
#ifdef RCC_EXPERIMENTS
  // slots 16..23 <- the category accumulators of the solver's inner steps
  if (g_grid_trace && threadIdx.x < 8) g_grid_trace[(size_t)f * 24 + 16 + threadIdx.x] = ((long long*)(ws + 308))[threadIdx.x];
#endif
}
__global__ __launch_bounds__(64) void k_grid_pnp(int w, int h, const rcc_cand* __restrict__ kept, const double* __restrict__ kept_xy,
                                                 int cols, int rows, rcc_frame_corners* __restrict__ fc,
                                                 const double* __restrict__ board_obj, double square, int board_id, int reference_mode,
                                                 rcc_cam cam, rcc_detection* __restrict__ det, int32_t* __restrict__ ndet)
{
  grid_pnp_frame(w, h, kept, kept_xy, cols, rows, fc, board_obj, square, board_id, reference_mode, cam, det, ndet);
}
#ifdef RCC_EXPERIMENTS
// experiment (rcc_set_tail_overlap): a marker the tail stream runs in front of the lattice + pose kernel (rcc_api.hip)
__global__ void k_marker() {}
hipError_t rcc_launch_marker(hipStream_t s) { hipLaunchKernelGGL(k_marker, dim3(1), dim3(64), 0, s); return hipGetLastError(); }
#endif

__global__ __launch_bounds__(64) void k_pnp_generic_wave(const double* __restrict__ obj, const double* __restrict__ img,
                                                         const int32_t* __restrict__ off, const int32_t* __restrict__ npts,
                                                         int ntargets, rcc_cam cam, double* __restrict__ rvec,
                                                         double* __restrict__ tvec, double* __restrict__ rms,
                                                         int32_t* __restrict__ status, int32_t* __restrict__ iters)
{
  const int t = blockIdx.x;
  const int lane = threadIdx.x;
  rccpnp::Pts p{ obj + 3 * (size_t)off[t], img + 2 * (size_t)off[t], npts[t] };
  __shared__ double ws[rccpnp::PNP_WS];
  RCC_GRAM_STAGE(gst);
  rccpnp::WavePar par{ lane, (unsigned)(uintptr_t)ws, gram_lds(gst) };
  double r[3], tv[3], e = 0.0;
  int it = 0;
  int st = rccpnp::solve_pnp(par, p, to_cam(cam), cam.model, r, tv, &e, &it);
  if (lane == 0) {
    for (int k = 0; k < 3; ++k) { rvec[3 * t + k] = r[k]; tvec[3 * t + k] = tv[k]; }
    if (rms) rms[t] = e;
    if (status) status[t] = st;
    if (iters) iters[t] = it;
  }
}

// test tap: intermediates of one solve, one thread.  out = H[9], prm_init[6], A[36], g[6], S, status
__global__ void k_pnp_probe(const double* obj, const double* img, int n, rcc_cam cam, double* out)
{
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  rccpnp::Pts p{ obj, img, n };
  rccpnp::Cam cm = to_cam(cam);
  const bool has_dist = cam.model == RCC_DIST_PLUMB_BOB;
  if (!has_dist) for (int i = 0; i < 5; ++i) cm.k[i] = 0.0;
  double wsl[rccpnp::PNP_WS];
  rccpnp::SerialPar par{ wsl };
  double Rt[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }, Tt[3] = { 0, 0, 0 };
  double H[9];
  int ok = rccpnp::find_homography(par, p, Rt, Tt, cm, has_dist, H);
  for (int i = 0; i < 9; ++i) out[i] = H[i];
  double prm[6];
  int st = rccpnp::pose_init(par, p, cm, has_dist, prm);
  for (int i = 0; i < 6; ++i) out[9 + i] = prm[i];
  double A[36], g[6];
  double S = rccpnp::pose_accumulate(par, prm, p, cm, A, g);
  for (int i = 0; i < 36; ++i) out[15 + i] = A[i];
  for (int i = 0; i < 6; ++i) out[51 + i] = g[i];
  out[57] = S;
  out[58] = (double)(st * 10 + ok);
}

hipError_t rcc_launch_pnp_probe(const double* d_obj, const double* d_img, int n, rcc_cam cam, double* d_out, hipStream_t s)
{
  hipLaunchKernelGGL(k_pnp_probe, dim3(1), dim3(64), 0, s, d_obj, d_img, n, cam, d_out);
  return hipGetLastError();
}

__global__ void k_rodrigues(int dir, const double* __restrict__ in, int n, double* __restrict__ out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (dir == 0) {
    double r[3] = { in[3 * i], in[3 * i + 1], in[3 * i + 2] }, R[9];
    rccpnp::rodrigues_v2m(r, R, nullptr);
    for (int k = 0; k < 9; ++k) out[9 * i + k] = R[k];
  } else {
    double R[9], r[3];
    for (int k = 0; k < 9; ++k) R[k] = in[9 * i + k];
    rccpnp::rodrigues_m2v(R, r);
    for (int k = 0; k < 3; ++k) out[3 * i + k] = r[k];
  }
}

hipError_t rcc_launch_pnp_generic(rcc_handle* h, const double* d_obj, const double* d_img,
                                  const int32_t* d_off, const int32_t* d_npts, int ntargets,
                                  rcc_cam cam, double* d_rvec, double* d_tvec, double* d_rms,
                                  int32_t* d_status, int32_t* d_iters, hipStream_t s)
{
  if (ntargets <= 0) return hipSuccess;
  if (h->pnp_wave_hint)   // many points per target: one wavefront each
    hipLaunchKernelGGL(k_pnp_generic_wave, dim3(ntargets), dim3(64), 0, s, d_obj, d_img, d_off, d_npts,
                       ntargets, cam, d_rvec, d_tvec, d_rms, d_status, d_iters);
  else
    hipLaunchKernelGGL(k_pnp_generic, dim3((ntargets + 63) / 64), dim3(64), 0, s, d_obj, d_img, d_off, d_npts,
                       ntargets, cam, d_rvec, d_tvec, d_rms, d_status, d_iters);
  return hipGetLastError();
}

hipError_t rcc_launch_pnp_board(rcc_handle* h, int nframes, hipStream_t s)
{
  const rcc_config& c = h->cfg;
  if (nframes <= 0) return hipSuccess;
  rcc_cam cam;
  cam.fx = c.K[0]; cam.cx = c.K[2]; cam.fy = c.K[4]; cam.cy = c.K[5];
  for (int i = 0; i < 8; ++i) cam.D[i] = c.D[i];
  cam.model = h->undist ? RCC_DIST_NONE : c.dist_model;   // undistorted image: solve with D = 0
  cam.solver = h->pnp_solver;
  if (c.board_cols * c.board_rows > 8 && h->pnp_variant != 0)
    hipLaunchKernelGGL(k_pnp_board_wave, dim3(nframes), dim3(64), 0, s, h->d_fc, nframes, h->d_board_obj,
                       c.board_cols, c.board_rows, c.board_square, c.board_id, c.reference_mode, cam,
                       h->d_img_scratch, h->d_det, h->d_ndet);
  else
    hipLaunchKernelGGL(k_pnp_board, dim3((nframes + 63) / 64), dim3(64), 0, s, h->d_fc, nframes, h->d_board_obj,
                       c.board_cols, c.board_rows, c.board_square, c.board_id, c.reference_mode, cam,
                       h->d_img_scratch, h->d_det, h->d_ndet);
  return hipGetLastError();
}

// validation, then lattice + board pose in one launch (wave per frame); same outputs as rcc_launch_grid followed by rcc_launch_pnp_board
hipError_t rcc_launch_grid_pnp(rcc_handle* h, const uint8_t* d_grey, const uint8_t* d_bin, int nframes, hipStream_t s)
{
  if (nframes <= 0) return hipSuccess;
  hipError_t e = rcc_launch_validate(h, d_grey, d_bin, nframes, s);
  if (e != hipSuccess) return e;
  return rcc_launch_grid_pnp_only(h, nframes, s, 0);
}
// the lattice + pose kernel alone (the validated lists are in place); lean: the 128-register build (experiments library)
hipError_t rcc_launch_grid_pnp_only(rcc_handle* h, int nframes, hipStream_t s, int lean)
{
  const rcc_config& c = h->cfg;
  if (nframes <= 0) return hipSuccess;
  rcc_cam cam;
  cam.fx = c.K[0]; cam.cx = c.K[2]; cam.fy = c.K[4]; cam.cy = c.K[5];
  for (int i = 0; i < 8; ++i) cam.D[i] = c.D[i];
  cam.model = h->undist ? RCC_DIST_NONE : c.dist_model;
  cam.solver = h->pnp_solver;
  (void)lean;
  hipLaunchKernelGGL(k_grid_pnp, dim3(nframes), dim3(64), 0, s, c.width, c.height, h->d_kept, h->d_kept_xy,
                     c.board_cols, c.board_rows, h->d_fc, h->d_board_obj, c.board_square, c.board_id,
                     c.reference_mode, cam, h->d_det, h->d_ndet);
  return hipGetLastError();
}
bool rcc_grid_pnp_applicable(const rcc_handle* h)
{
  const rcc_config& c = h->cfg;
  return c.target_kind == RCC_TARGET_CHECKERBOARD && c.board_cols * c.board_rows > 8 && h->pnp_variant != 0;
}

hipError_t rcc_launch_rodrigues(int dir, const double* d_in, int n, double* d_out, hipStream_t s)
{
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_rodrigues, dim3((n + 63) / 64), dim3(64), 0, s, dir, d_in, n, d_out);
  return hipGetLastError();
}
