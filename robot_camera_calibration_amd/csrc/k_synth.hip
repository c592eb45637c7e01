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
  int w, h, stride, nch, rgb;      // rgb: RCC_PIX_RGB8 -- the three values of a pixel are stored in the opposite order
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

// the s x s samples of pixel (u, v) summed per channel: integers (class colour + tint each), exact in double
__device__ __forceinline__ void sample_pixel(const synth_args& a, const double* __restrict__ Hi, int u, int v, double acc[3])
{
  const double hx = 0.5 * a.nsx, hy = 0.5 * a.nsy, mg = (double)a.margin;
  const bool fid = a.fid_gx > 0 && a.fid_gy > 0;
  const double fhx = 0.5 * (a.fid_gx * a.pitch - (a.pitch - a.tag)) + 0.5 * a.tag;
  const double fhy = 0.5 * (a.fid_gy * a.pitch - (a.pitch - a.tag)) + 0.5 * a.tag;
  const int tint[3][3] = { { 10, 0, -10 }, { -8, 0, -3 }, { 4, 0, 2 } };
  acc[0] = acc[1] = acc[2] = 0.0;
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
}

// sensor: noise, rounding, clamping of one pixel's channel values (val[c] before the noise)
__device__ __forceinline__ void store_pixel(const synth_args& a, int f, int u, int v, const double val3[3], uint8_t* __restrict__ frames)
{
  const uint64_t key = splitmix64(a.seed + (uint64_t)(a.first_index + f));
  const size_t pix = (size_t)v * a.w + u;
  uint8_t* dst = frames + (size_t)f * a.frame_bytes + (size_t)v * a.stride + (size_t)u * a.nch;
  for (int c = 0; c < a.nch; ++c) {
    double val = val3[a.nch == 3 ? c : 1];
    val += a.sigma * hash_gauss(key, (uint64_t)pix * 3u + (uint64_t)c);
    double rr = rint(val);
    int iv = rr < 0.0 ? 0 : (rr > 255.0 ? 255 : (int)rr);
    dst[(a.rgb && a.nch == 3) ? 2 - c : c] = (uint8_t)iv;       // c counts blue, green, red
  }
}

// the ideal camera (all optics parameters zero): mean of the samples + noise, one pass
__global__ __launch_bounds__(256) void k_synth_render(synth_args a, const double* __restrict__ hinv, uint8_t* __restrict__ frames)
{
  const int u = blockIdx.x * 64 + threadIdx.x;
  const int v = blockIdx.y * 4 + threadIdx.y;
  const int f = blockIdx.z;
  if (u >= a.w || v >= a.h) return;
  double acc[3];
  sample_pixel(a, hinv + 9 * (size_t)f, u, v, acc);
  double val[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) val[c] = acc[c] / (double)(a.ss * a.ss);
  store_pixel(a, f, u, v, val, frames);
}

// ---- optics (rcc_synth_params, ABI 2): the supersampled sums as integers -> separable integer blur -> integer gain -> sensor.
// pass 1: A[f][v][u][3] = the sums
__global__ __launch_bounds__(256) void k_synth_sums(synth_args a, const double* __restrict__ hinv, int32_t* __restrict__ A)
{
  const int u = blockIdx.x * 64 + threadIdx.x;
  const int v = blockIdx.y * 4 + threadIdx.y;
  const int f = blockIdx.z;
  if (u >= a.w || v >= a.h) return;
  double acc[3];
  sample_pixel(a, hinv + 9 * (size_t)f, u, v, acc);
  int32_t* o = A + (((size_t)f * a.h + v) * a.w + u) * 3;
  o[0] = (int32_t)acc[0]; o[1] = (int32_t)acc[1]; o[2] = (int32_t)acc[2];
}
struct synth_optics {
  int taps[RCC_SYNTH_BLUR_TAPS];
  int shade_x, shade_y, vignette;
};
// pass 2: along the rows (weights in 1/256, columns clamped at the border)
__global__ __launch_bounds__(256) void k_synth_blur_rows(int w, int h, synth_optics o, const int32_t* __restrict__ A, int32_t* __restrict__ T)
{
  const int u = blockIdx.x * 64 + threadIdx.x;
  const int v = blockIdx.y * 4 + threadIdx.y;
  const int f = blockIdx.z;
  if (u >= w || v >= h) return;
  const int32_t* row = A + ((size_t)f * h + v) * (size_t)w * 3;
  int32_t r0 = 0, r1 = 0, r2 = 0;
  for (int i = -(RCC_SYNTH_BLUR_TAPS - 1); i < RCC_SYNTH_BLUR_TAPS; ++i) {
    const int wi = o.taps[i < 0 ? -i : i];
    if (!wi) continue;
    const int uu = min(max(u + i, 0), w - 1);
    r0 += wi * row[uu * 3]; r1 += wi * row[uu * 3 + 1]; r2 += wi * row[uu * 3 + 2];
  }
  int32_t* t = T + (((size_t)f * h + v) * w + u) * 3;
  t[0] = r0; t[1] = r1; t[2] = r2;
}
// pass 3: along the columns (1/256 again, rows clamped), the gain in 1/4096, ONE division, noise, rounding
__global__ __launch_bounds__(256) void k_synth_finish(synth_args a, synth_optics o, const int32_t* __restrict__ T, int frame0, uint8_t* __restrict__ frames)
{
  const int u = blockIdx.x * 64 + threadIdx.x;
  const int v = blockIdx.y * 4 + threadIdx.y;
  const int f = blockIdx.z;
  if (u >= a.w || v >= a.h) return;
  long long b0 = 0, b1 = 0, b2 = 0;
  for (int j = -(RCC_SYNTH_BLUR_TAPS - 1); j < RCC_SYNTH_BLUR_TAPS; ++j) {
    const int wj = o.taps[j < 0 ? -j : j];
    if (!wj) continue;
    const int vv = min(max(v + j, 0), a.h - 1);
    const int32_t* t = T + (((size_t)f * a.h + vv) * a.w + u) * 3;
    b0 += (long long)wj * t[0]; b1 += (long long)wj * t[1]; b2 += (long long)wj * t[2];
  }
  const long long X = 2 * (long long)u - (a.w - 1), Y = 2 * (long long)v - (a.h - 1);
  const long long W1 = a.w > 1 ? a.w - 1 : 1, H1 = a.h > 1 ? a.h - 1 : 1;
  const long long lin = 4096 + (4096 * (long long)o.shade_x * X) / (1000 * W1) + (4096 * (long long)o.shade_y * Y) / (1000 * H1);
  const long long r2 = X * X + Y * Y, R2 = (long long)(a.w - 1) * (a.w - 1) + (long long)(a.h - 1) * (a.h - 1);
  const long long vig = 4096 - (4096 * (long long)o.vignette * r2) / (1000 * (R2 > 0 ? R2 : 1));
  const long long g = (lin * vig) >> 12;
  const double scale = 256.0 * 256.0 * 4096.0 * (double)(a.ss * a.ss);
  double val[3] = { (double)(b0 * g) / scale, (double)(b1 * g) / scale, (double)(b2 * g) / scale };
  store_pixel(a, frame0 + f, u, v, val, frames);
}

// 0 = all off (the ideal camera), 1 = on and well-formed, -1 = malformed (include/rcc.h: taps >= 0, centre + 2 x the rest = 256)
int rcc_synth_optics(const rcc_synth_params* sp, int taps[RCC_SYNTH_BLUR_TAPS])
{
  int any = 0, sum = 0;
  for (int k = 0; k < RCC_SYNTH_BLUR_TAPS; ++k) {
    taps[k] = sp->blur_taps[k];
    if (taps[k] < 0) return -1;
    any |= taps[k] != 0;
    sum += (k ? 2 : 1) * taps[k];
  }
  if (any && sum != 256) return -1;
  if (!any) taps[0] = 256;                    // shading without blur: the identity filter
  if (sp->vignette_permille < 0 || sp->vignette_permille > 1000) return -1;
  if (abs(sp->shade_x_permille) + abs(sp->shade_y_permille) > 1000) return -1;
  return (any || sp->shade_x_permille || sp->shade_y_permille || sp->vignette_permille) ? 1 : 0;
}

hipError_t rcc_launch_synth(rcc_handle* h, const rcc_synth_params* sp, const double* d_hinv,
                            int nframes, int first_index, uint8_t* d_frames, hipStream_t s)
{
  const rcc_config& c = h->cfg;
  if (nframes <= 0) return hipSuccess;
  synth_args a;
  a.w = c.width; a.h = c.height; a.stride = c.stride_bytes; a.nch = (c.pixfmt == RCC_PIX_MONO8) ? 1 : 3;
  a.rgb = (c.pixfmt == RCC_PIX_RGB8) ? 1 : 0;
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
  synth_optics o;
  const int optics = rcc_synth_optics(sp, o.taps);
  if (optics < 0) return hipErrorInvalidValue;
  o.shade_x = sp->shade_x_permille; o.shade_y = sp->shade_y_permille; o.vignette = sp->vignette_permille;
  dim3 block(64, 4);
  if (!optics) {
    dim3 grid((c.width + 63) / 64, (c.height + 3) / 4, nframes);
    hipLaunchKernelGGL(k_synth_render, grid, block, 0, s, a, d_hinv, d_frames);
    return hipGetLastError();
  }
  // two int32 x 3 images per frame in flight: at most ~2 GiB of scratch, the batch goes through in groups of frames
  const size_t per_frame = (size_t)c.width * c.height * 3 * sizeof(int32_t);
  int group = (int)(((size_t)1 << 30) / per_frame);
  if (group < 1) group = 1;
  if (group > nframes) group = nframes;
  const size_t need = 2 * per_frame * (size_t)group;
  if (need > h->synth_tmp_bytes) {
    if (h->d_synth_tmp) (void)hipFree(h->d_synth_tmp);
    h->d_synth_tmp = nullptr; h->synth_tmp_bytes = 0;
    hipError_t e = hipMalloc((void**)&h->d_synth_tmp, need);
    if (e != hipSuccess) return e;
    h->synth_tmp_bytes = need;
  }
  int32_t* A = (int32_t*)h->d_synth_tmp;
  int32_t* T = A + (size_t)group * c.width * c.height * 3;
  for (int f0 = 0; f0 < nframes; f0 += group) {
    const int n = (nframes - f0 < group) ? nframes - f0 : group;
    dim3 grid((c.width + 63) / 64, (c.height + 3) / 4, n);
    synth_args b = a;
    b.first_index = a.first_index;      // frame f of this group is batch frame f0 + f: the sums read pose f0 + f, the sensor's key is first_index + f0 + f
    hipLaunchKernelGGL(k_synth_sums, grid, block, 0, s, b, d_hinv + 9 * (size_t)f0, A);
    hipLaunchKernelGGL(k_synth_blur_rows, grid, block, 0, s, c.width, c.height, o, A, T);
    hipLaunchKernelGGL(k_synth_finish, grid, block, 0, s, b, o, T, f0, d_frames);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}
