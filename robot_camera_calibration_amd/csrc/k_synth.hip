// k_synth.hip -- synthetic camera: renders checkerboard frames on the device.
//
// Stands where the reference's simulator was meant to grow a camera: rviz_simulator publishes one
// interactive cube and no images (rviz_simulator/src/simulate.cpp:44-68); the "Camera class
// defined in camera.h" of rviz_simulator/include/rviz_simulator/target.h:40 does not exist.
// BASELINE.json's workloads are synthetic checkerboard frames, so the generator is the build's
// own: pinhole + plumb-bob / fisheye camera, (cols+1)x(rows+1)-square
// board (8x6 inner corners of 0.108 m, real_preprocessing/README.md:57) with a white quiet zone,
// s x s supersampling, hash noise.  Not on the timed path; workload generation only.
#include "rcc_internal.h"

__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__device__ __forceinline__ double hash_gauss(uint64_t key, uint64_t idx)
{
  uint64_t h = splitmix64(key ^ (idx * 0xD1342543DE82EF95ull));
  int s = (int)(h & 0xFFFF) + (int)((h >> 16) & 0xFFFF) + (int)((h >> 32) & 0xFFFF) + (int)((h >> 48) & 0xFFFF);
  return (double)(s - 131070) / 37837.2;
}

__device__ __forceinline__ bool undistort_norm(int model, const double* D, double xd, double yd, double& x, double& y)
{
  if (model == RCC_DIST_PLUMB_BOB) {
    double px = xd, py = yd;
    for (int it = 0; it < 20; ++it) {
      double r2 = px * px + py * py;
      double ic = 1.0 / (1.0 + ((D[4] * r2 + D[1]) * r2 + D[0]) * r2);
      double dx = 2.0 * D[2] * px * py + D[3] * (r2 + 2.0 * px * px);
      double dy = D[2] * (r2 + 2.0 * py * py) + 2.0 * D[3] * px * py;
      px = (xd - dx) * ic;
      py = (yd - dy) * ic;
    }
    x = px; y = py;
    return true;
  }
  if (model == RCC_DIST_FISHEYE) {
    double thd = sqrt(xd * xd + yd * yd);
    if (thd < 1e-8) { x = xd; y = yd; return true; }
    double th = thd;
    for (int it = 0; it < 10; ++it) {
      double t2 = th * th, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
      double f = th * (1.0 + D[0] * t2 + D[1] * t4 + D[2] * t6 + D[3] * t8) - thd;
      double fp = 1.0 + 3.0 * D[0] * t2 + 5.0 * D[1] * t4 + 7.0 * D[2] * t6 + 9.0 * D[3] * t8;
      th = th - f / fp;
    }
    if (!(th > 0.0) || th >= 1.5) return false;
    double sc = tan(th) / thd;
    x = xd * sc; y = yd * sc;
    return true;
  }
  x = xd; y = yd;
  return true;
}

struct synth_args {
  int w, h, stride, nch;
  int64_t frame_bytes;
  rcc_cam cam;
  int ss, nsx, nsy, margin;
  double sq, sigma;
  int base[3];
  uint64_t seed;
  int first_index;
  int fid_gx, fid_gy, family_n;
  double tag, pitch;
  const uint64_t* codes;
};

__global__ __launch_bounds__(256) void k_synth_render(synth_args a, const double* __restrict__ hinv, uint8_t* __restrict__ frames)
{
  const int u = blockIdx.x * 64 + threadIdx.x;
  const int v = blockIdx.y * 4 + threadIdx.y;
  const int f = blockIdx.z;
  if (u >= a.w || v >= a.h) return;
  const double* Hi = hinv + 9 * (size_t)f;
  const double hx = 0.5 * a.nsx, hy = 0.5 * a.nsy, mg = (double)a.margin;
  const bool fid = a.fid_gx > 0 && a.fid_gy > 0;
  const double fhx = 0.5 * (a.fid_gx * a.pitch - (a.pitch - a.tag)) + 0.5 * a.tag;
  const double fhy = 0.5 * (a.fid_gy * a.pitch - (a.pitch - a.tag)) + 0.5 * a.tag;
  const int tint[3][3] = { { 10, 0, -10 }, { -8, 0, -3 }, { 4, 0, 2 } };
  double acc[3] = { 0, 0, 0 };
  for (int sy = 0; sy < a.ss; ++sy)
    for (int sx = 0; sx < a.ss; ++sx) {
      double us = (double)u + ((double)sx + 0.5) / a.ss - 0.5;
      double vs = (double)v + ((double)sy + 0.5) / a.ss - 0.5;
      double xd = (us - a.cam.cx) / a.cam.fx, yd = (vs - a.cam.cy) / a.cam.fy, x, y;
      int cls = 0;
      if (undistort_norm(a.cam.model, a.cam.D, xd, yd, x, y)) {
        double q0 = Hi[0] * x + Hi[1] * y + Hi[2];
        double q1 = Hi[3] * x + Hi[4] * y + Hi[5];
        double q2 = Hi[6] * x + Hi[7] * y + Hi[8];
        if (q2 > 0.0 && fid) {
          const double Xm = q0 / q2, Ym = q1 / q2;
          if (fabs(Xm) < fhx && fabs(Ym) < fhy) {
            cls = 1;
            const double gx = (Xm + fhx - 0.5 * a.tag) / a.pitch, gy = (fhy - 0.5 * a.tag - Ym) / a.pitch;
            const int ti = (int)floor(gx), tj = (int)floor(gy);
            if (ti >= 0 && tj >= 0 && ti < a.fid_gx && tj < a.fid_gy) {
              const double uu = (gx - ti) * a.pitch / a.tag * 8.0, vv = (gy - tj) * a.pitch / a.tag * 8.0;
              if (uu < 8.0 && vv < 8.0) {
                const int cu = (int)floor(uu), cv = (int)floor(vv);
                if (cu == 0 || cv == 0 || cu == 7 || cv == 7) cls = 2;
                else {
                  const uint64_t code = a.codes[(tj * a.fid_gx + ti) % a.family_n];
                  cls = ((code >> (35 - ((cv - 1) * 6 + (cu - 1)))) & 1u) ? 1 : 2;
                }
              }
            }
          }
        } else if (q2 > 0.0) {
          double X = q0 / q2 / a.sq, Y = q1 / q2 / a.sq;
          if (fabs(X) < hx + mg && fabs(Y) < hy + mg) {
            cls = 1;
            if (fabs(X) < hx && fabs(Y) < hy) {
              int i = (int)floor(X + hx), j = (int)floor(Y + hy);
              cls = ((i + j) & 1) ? 1 : 2;
            }
          }
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[c] += (double)(a.base[cls] + tint[cls][c]);
    }
  const uint64_t key = splitmix64(a.seed + (uint64_t)(a.first_index + f));
  const size_t pix = (size_t)v * a.w + u;
  uint8_t* dst = frames + (size_t)f * a.frame_bytes + (size_t)v * a.stride + (size_t)u * a.nch;
  for (int c = 0; c < a.nch; ++c) {
    double val = acc[a.nch == 3 ? c : 1] / (double)(a.ss * a.ss);
    val += a.sigma * hash_gauss(key, (uint64_t)pix * 3u + (uint64_t)c);
    double rr = rint(val);
    int iv = rr < 0.0 ? 0 : (rr > 255.0 ? 255 : (int)rr);
    dst[c] = (uint8_t)iv;
  }
}

hipError_t rcc_launch_synth(rcc_handle* h, const rcc_synth_params* sp, const double* d_hinv,
                            int nframes, int first_index, uint8_t* d_frames, hipStream_t s)
{
  const rcc_config& c = h->cfg;
  if (nframes <= 0) return hipSuccess;
  synth_args a;
  a.w = c.width; a.h = c.height; a.stride = c.stride_bytes; a.nch = (c.pixfmt == RCC_PIX_BGR8) ? 3 : 1;
  a.frame_bytes = c.frame_bytes;
  a.cam.fx = c.K[0]; a.cam.cx = c.K[2]; a.cam.fy = c.K[4]; a.cam.cy = c.K[5];
  for (int i = 0; i < 8; ++i) a.cam.D[i] = c.D[i];
  a.cam.model = c.dist_model;
  a.cam.solver = 0;
  a.ss = sp->supersample < 1 ? 1 : sp->supersample;
  a.nsx = sp->board_cols + 1; a.nsy = sp->board_rows + 1; a.margin = sp->margin_squares;
  a.sq = sp->board_square; a.sigma = sp->noise_sigma;
  a.base[0] = sp->background; a.base[1] = sp->white; a.base[2] = sp->black;
  a.seed = sp->seed;
  a.first_index = first_index;
  a.fid_gx = a.fid_gy = 0; a.family_n = 1; a.tag = 1.0; a.pitch = 1.5; a.codes = nullptr;
  if (sp->fid_grid_x > 0 && sp->fid_grid_y > 0 && h->d_family) {
    a.fid_gx = sp->fid_grid_x; a.fid_gy = sp->fid_grid_y; a.family_n = c.family_n;
    a.tag = c.tag_size; a.pitch = c.tag_size * (1.0 + 0.001 * sp->fid_gap_permille); a.codes = h->d_family;
  }
  dim3 grid((c.width + 63) / 64, (c.height + 3) / 4, nframes), block(64, 4);
  hipLaunchKernelGGL(k_synth_render, grid, block, 0, s, a, d_hinv, d_frames);
  return hipGetLastError();
}
