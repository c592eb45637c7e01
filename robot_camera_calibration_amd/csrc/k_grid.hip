// k_grid.hip -- stages a4.3 (validation of refined corners: X-junction ring test on the
// threshold map + de-duplication) and a6 (board indexing by seeded lattice growth on integer
// positions, un-shear, orientation).  One 256-thread block per frame: the validation is one candidate per
// thread; the lattice stage then runs in the block's first wavefront with the frame's points in registers
// (grid_frame.h): control flow is wave-uniform, the searches (nearest free point, seed ranking, bounding
// boxes) run across lanes and finish with a min/max butterfly whose key carries the index, so ties resolve
// to the smaller index exactly as the specification's serial scans do.
//
// Consumer-side contract: per detection the reference reads id[0], size[0] and four pixel corners
// (real_preprocessing/src/corner_detections.cpp:48-54); corner order and object frame follow
// real_preprocessing/src/camera_pose.cpp:152-161.  Definitions: DESIGN.md section 3 (a4.3, a6).
#include "grid_frame.h"

__global__ __launch_bounds__(256) void k_validate(const uint8_t* __restrict__ bin, const uint8_t* __restrict__ grey,
                                                 const uint8_t* __restrict__ thr, int nbands, int w, int h,
                                                 const rcc_cand* __restrict__ pre, const int32_t* __restrict__ npre,
                                                 const double* __restrict__ pre_xy, int pre_stride, int max_kept, int xj_check, int min_contrast, int dedupe_radius,
                                                 rcc_frame_corners* __restrict__ fc,
                                                 rcc_cand* __restrict__ kept_out, double* __restrict__ kept_xy_out)
{
  __shared__ valid_smem sm;
  validate_frame<256>(sm, blockIdx.x, threadIdx.x, bin, grey, thr, nbands, w, h, pre, npre, pre_xy, pre_stride, max_kept, xj_check, min_contrast, dedupe_radius, fc, kept_out, kept_xy_out);
}

__global__ __launch_bounds__(64) void k_grid_index(int w, int h, const rcc_cand* __restrict__ kept, const double* __restrict__ kept_xy,
                                                   int target_kind, int cols, int rows, rcc_frame_corners* __restrict__ fc)
{
  __shared__ grid_smem sm;
  index_frame(sm, blockIdx.x, threadIdx.x, w, h, kept, kept_xy, target_kind, cols, rows, fc);
}

// a4.3 for the frames of a batch: the handle's suppressed lists + refined positions -> kept lists, fc[].nkept
hipError_t rcc_launch_validate(rcc_handle* h, const uint8_t* d_grey, const uint8_t* d_bin, int nframes, hipStream_t s)
{
  const rcc_config& c = h->cfg;
  if (nframes <= 0) return hipSuccess;
  const int nbands = (c.width + RCC_BAND_W - 1) / RCC_BAND_W;
  hipLaunchKernelGGL(k_validate, dim3(nframes), dim3(256), 0, s, h->bin_from_thr ? nullptr : d_bin, d_grey,
                     h->bin_from_thr ? h->d_thr : nullptr, nbands, c.width, c.height,
                     h->d_pre, h->d_npre, h->d_pre_xy, h->kept_cap, c.max_kept < RCC_MAX_KEPT ? c.max_kept : RCC_MAX_KEPT, c.xj_check, c.thr_min_contrast, 2,
                     h->d_fc, h->d_kept, h->d_kept_xy);
  return hipGetLastError();
}

hipError_t rcc_launch_grid(rcc_handle* h, const uint8_t* d_grey, const uint8_t* d_bin, int nframes, hipStream_t s)
{
  const rcc_config& c = h->cfg;
  if (nframes <= 0) return hipSuccess;
  hipError_t e = rcc_launch_validate(h, d_grey, d_bin, nframes, s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_grid_index, dim3(nframes), dim3(64), 0, s, c.width, c.height, h->d_kept, h->d_kept_xy, c.target_kind, c.board_cols,
                     c.board_rows, h->d_fc);
  return hipGetLastError();
}

// the full binary image from the grey image and the compact threshold map (debug / parity taps only)
__global__ __launch_bounds__(256) void k_expand_bin(const uint8_t* __restrict__ grey, const uint8_t* __restrict__ thr, int nbands,
                                                    int w, int h, uint8_t* __restrict__ bin)
{
  const int f = blockIdx.z, y = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
  if (x >= w) return;
  BinSrc b;
  b.bin = nullptr; b.grey = grey + (size_t)f * w * h; b.thr = thr + (size_t)f * nbands * (h >> 2) * RCC_THR_PITCH; b.w = w; b.th = h >> 2;
  bin[(size_t)f * w * h + (size_t)y * w + x] = (uint8_t)b.at(x, y);
}
hipError_t rcc_launch_expand_bin(rcc_handle* h, const uint8_t* d_grey, int nframes, uint8_t* d_bin, hipStream_t s)
{
  const rcc_config& c = h->cfg;
  if (nframes <= 0) return hipSuccess;
  const int nbands = (c.width + RCC_BAND_W - 1) / RCC_BAND_W;
  hipLaunchKernelGGL(k_expand_bin, dim3((c.width + 255) / 256, c.height, nframes), dim3(256), 0, s, d_grey, h->d_thr, nbands, c.width, c.height, d_bin);
  return hipGetLastError();
}
