// k_pnp.hip -- stage a7/a8 kernels: per-target pose (solvePnP ITERATIVE form) and Rodrigues.
//
// Drop-in target: cv::solvePnP(...) at real_preprocessing/src/camera_pose.cpp:163 and
// cv::Rodrigues at :164 (and :93,:116; opt_visualization.cpp:36).  Arithmetic in pnp_core.h.
//
// Mapping: one lane per target.  A 4-point fiducial solve is ~1e5 dependent fp64 operations with
// 6x6/9x9 eigen-decompositions in the middle; it does not vectorise across points, so targets are
// the parallel axis: 64 targets per wavefront, thousands of targets per batch.  The result record
// layout is the C ABI's rcc_detection (what corner_detections.cpp:46-56 reads, plus the pose).
#include "rcc_internal.h"
#include "pnp_core.h"

static __device__ __forceinline__ rccpnp::Cam to_cam(const rcc_cam& c)
{
  rccpnp::Cam k;
  k.fx = c.fx; k.fy = c.fy; k.cx = c.cx; k.cy = c.cy;
  for (int i = 0; i < 5; ++i) k.k[i] = c.D[i];
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
  int it = 0;
  int st = rccpnp::solve_pnp(p, to_cam(cam), cam.model, r, tv, &e, &it);
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
  d.pnp_status = rccpnp::solve_pnp(p, to_cam(cam), cam.model, d.rvec, d.tvec, &d.rms, &it);
  d.pnp_iters = it;
  det[f] = d;
  ndet[f] = 1;
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
  hipLaunchKernelGGL(k_pnp_board, dim3((nframes + 63) / 64), dim3(64), 0, s, h->d_fc, nframes, h->d_board_obj,
                     c.board_cols, c.board_rows, c.board_square, c.board_id, c.reference_mode, cam,
                     h->d_img_scratch, h->d_det, h->d_ndet);
  return hipGetLastError();
}

hipError_t rcc_launch_rodrigues(int dir, const double* d_in, int n, double* d_out, hipStream_t s)
{
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_rodrigues, dim3((n + 63) / 64), dim3(64), 0, s, dir, d_in, n, d_out);
  return hipGetLastError();
}
