/*
 * orc_synth.c -- CPU version of the synthetic camera [B].  TEST INFRASTRUCTURE.
 *
 * The reference renders no images: rviz_simulator publishes one interactive cube
 * (rviz_simulator/src/simulate.cpp:44-68) and the "Camera class defined in camera.h" its header
 * mentions (rviz_simulator/include/rviz_simulator/target.h:40) is absent.  BASELINE.json's
 * configs need checkerboard frames, so the build supplies a deterministic generator: a pinhole +
 * plumb-bob / fisheye camera looking at an (cols+1) x (rows+1)-square board (8x6 inner corners,
 * 0.108 m squares: real_preprocessing/README.md:57) with a white quiet zone, s x s supersampled,
 * plus hash-based approximately Gaussian noise.  Parameters as SURVEY.md section 8(d).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "orc.h"

static uint64_t splitmix64(uint64_t x)
{
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

/* sum of four 16-bit uniforms, centred and scaled to unit variance */
static double hash_gauss(uint64_t key, uint64_t idx)
{
  uint64_t h = splitmix64(key ^ (idx * 0xD1342543DE82EF95ull));
  int s = (int)(h & 0xFFFF) + (int)((h >> 16) & 0xFFFF) + (int)((h >> 32) & 0xFFFF) + (int)((h >> 48) & 0xFFFF);
  return (double)(s - 131070) / 37837.2;
}

static void inv3(const double* M, double* I)
{
  double d = M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
  double s = 1.0 / d;
  I[0] = (M[4] * M[8] - M[5] * M[7]) * s; I[1] = (M[2] * M[7] - M[1] * M[8]) * s; I[2] = (M[1] * M[5] - M[2] * M[4]) * s;
  I[3] = (M[5] * M[6] - M[3] * M[8]) * s; I[4] = (M[0] * M[8] - M[2] * M[6]) * s; I[5] = (M[2] * M[3] - M[0] * M[5]) * s;
  I[6] = (M[3] * M[7] - M[4] * M[6]) * s; I[7] = (M[1] * M[6] - M[0] * M[7]) * s; I[8] = (M[0] * M[4] - M[1] * M[3]) * s;
}

/* distorted normalised -> ideal normalised; returns 0 when the ray is not representable */
static int undistort_norm(int model, const double* D, double xd, double yd, double* x, double* y)
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
    *x = px; *y = py;
    return 1;
  }
  if (model == RCC_DIST_FISHEYE) {
    double thd = sqrt(xd * xd + yd * yd);
    if (thd < 1e-8) { *x = xd; *y = yd; return 1; }
    double th = thd;
    for (int it = 0; it < 10; ++it) {
      double t2 = th * th, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
      double f = th * (1.0 + D[0] * t2 + D[1] * t4 + D[2] * t6 + D[3] * t8) - thd;
      double fp = 1.0 + 3.0 * D[0] * t2 + 5.0 * D[1] * t4 + 7.0 * D[2] * t6 + 9.0 * D[3] * t8;
      th = th - f / fp;
    }
    if (!(th > 0.0) || th >= 1.5) return 0;
    double sc = tan(th) / thd;
    *x = xd * sc; *y = yd * sc;
    return 1;
  }
  *x = xd; *y = yd;
  return 1;
}

/* class of one sample of one pixel: 0 background, 1 white, 2 black */
typedef struct synth_scene {
  const rcc_config* cfg; const rcc_synth_params* sp;
  double fx, fy, cx, cy, Hi[9];
  int ss, fid, nsx, nsy;
  double sq, tag, pitch, fhx, fhy, hx, hy, mg;
} synth_scene;

/* the s x s samples of pixel (u, v) summed per channel: integers (class colour + tint each), exact in double */
static void sample_pixel(const synth_scene* S, int u, int v, double acc[3])
{
  const rcc_config* cfg = S->cfg; const rcc_synth_params* sp = S->sp;
  const int ss = S->ss;
  const double* Hi = S->Hi;
  /* class colours (B,G,R): slight tints so the grey conversion is exercised */
  const int base[3] = { sp->background, sp->white, sp->black };
  const int tint[3][3] = { { 10, 0, -10 }, { -8, 0, -3 }, { 4, 0, 2 } };
  acc[0] = acc[1] = acc[2] = 0.0;
  for (int sy = 0; sy < ss; ++sy)
    for (int sx = 0; sx < ss; ++sx) {
      double us = (double)u + ((double)sx + 0.5) / ss - 0.5;
      double vs = (double)v + ((double)sy + 0.5) / ss - 0.5;
      double xd = (us - S->cx) / S->fx, yd = (vs - S->cy) / S->fy, x, y;
      int cls = 0;
      if (undistort_norm(cfg->dist_model, cfg->D, xd, yd, &x, &y)) {
        double q0 = Hi[0] * x + Hi[1] * y + Hi[2];
        double q1 = Hi[3] * x + Hi[4] * y + Hi[5];
        double q2 = Hi[6] * x + Hi[7] * y + Hi[8];
        if (q2 > 0.0 && S->fid) {
          /* metres on the tag plane, x right, y up (object frame of camera_pose.cpp:158-161) */
          const double Xm = q0 / q2, Ym = q1 / q2;
          if (fabs(Xm) < S->fhx && fabs(Ym) < S->fhy) {
            cls = 1;
            const double gx = (Xm + S->fhx - 0.5 * S->tag) / S->pitch, gy = (S->fhy - 0.5 * S->tag - Ym) / S->pitch;   /* grid coords, row 0 on top */
            const int ti = (int)floor(gx), tj = (int)floor(gy);
            if (ti >= 0 && tj >= 0 && ti < sp->fid_grid_x && tj < sp->fid_grid_y) {
              const double cu_ = (gx - ti) * S->pitch / S->tag * 8.0, cv_ = (gy - tj) * S->pitch / S->tag * 8.0;   /* cells */
              if (cu_ < 8.0 && cv_ < 8.0) {
                const int cu = (int)floor(cu_), cv = (int)floor(cv_);
                if (cu == 0 || cv == 0 || cu == 7 || cv == 7) cls = 2;
                else {
                  const uint64_t code = cfg->family_codes[(tj * sp->fid_grid_x + ti) % cfg->family_n];
                  cls = ((code >> (35 - ((cv - 1) * 6 + (cu - 1)))) & 1u) ? 1 : 2;
                }
              }
            }
          }
        } else if (q2 > 0.0) {
          double X = q0 / q2 / S->sq, Y = q1 / q2 / S->sq;   /* in squares, origin at the centre */
          if (fabs(X) < S->hx + S->mg && fabs(Y) < S->hy + S->mg) {
            cls = 1;
            if (fabs(X) < S->hx && fabs(Y) < S->hy) {
              int i = (int)floor(X + S->hx), j = (int)floor(Y + S->hy);
              cls = ((i + j) & 1) ? 1 : 2;
            }
          }
        }
      }
      for (int c = 0; c < 3; ++c) acc[c] += (double)(base[cls] + tint[cls][c]);
    }
}

/* optics of rcc_synth_params (include/rcc.h, ABI 2): 0 = all off (the ideal camera), 1 = on and well-formed, -1 = malformed taps */
int orc_synth_optics(const rcc_synth_params* sp, int taps[RCC_SYNTH_BLUR_TAPS])
{
  int any = 0, sum = 0;
  for (int k = 0; k < RCC_SYNTH_BLUR_TAPS; ++k) {
    taps[k] = sp->blur_taps[k];
    if (taps[k] < 0) return -1;
    any |= taps[k] != 0;
    sum += (k ? 2 : 1) * taps[k];
  }
  if (any && sum != 256) return -1;
  if (!any) taps[0] = 256;                    /* shading without blur: the identity filter */
  if (sp->vignette_permille < 0 || sp->vignette_permille > 1000) return -1;
  if (abs(sp->shade_x_permille) + abs(sp->shade_y_permille) > 1000) return -1;
  return (any || sp->shade_x_permille || sp->shade_y_permille || sp->vignette_permille) ? 1 : 0;
}

/* gain of pixel (u, v) in 1/4096 (include/rcc.h): integer arithmetic, C truncation */
static int64_t shading_gain(const rcc_synth_params* sp, int w, int h, int u, int v)
{
  const int64_t X = 2 * (int64_t)u - (w - 1), Y = 2 * (int64_t)v - (h - 1);
  const int64_t W1 = w > 1 ? w - 1 : 1, H1 = h > 1 ? h - 1 : 1;
  const int64_t lin = 4096 + (4096 * (int64_t)sp->shade_x_permille * X) / (1000 * W1) + (4096 * (int64_t)sp->shade_y_permille * Y) / (1000 * H1);
  const int64_t r2 = X * X + Y * Y, R2 = (int64_t)(w - 1) * (w - 1) + (int64_t)(h - 1) * (h - 1);
  const int64_t vig = 4096 - (4096 * (int64_t)sp->vignette_permille * r2) / (1000 * (R2 > 0 ? R2 : 1));
  return (lin * vig) >> 12;
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

int orc_synth_render(const rcc_config* cfg, const rcc_synth_params* sp, const double pose[6],
                     int frame_index, uint8_t* out)
{
  const int w = cfg->width, h = cfg->height;
  synth_scene S;
  S.cfg = cfg; S.sp = sp;
  S.fx = cfg->K[0]; S.cx = cfg->K[2]; S.fy = cfg->K[4]; S.cy = cfg->K[5];
  double R[9];
  orc_rodrigues_v2m(pose, R, NULL);
  double H[9] = { R[0], R[1], pose[3], R[3], R[4], pose[4], R[6], R[7], pose[5] };
  inv3(H, S.Hi);
  S.ss = sp->supersample < 1 ? 1 : sp->supersample;
  S.sq = sp->board_square;
  S.fid = sp->fid_grid_x > 0 && sp->fid_grid_y > 0;       /* planar grid of fiducials instead of the board */
  S.tag = cfg->tag_size; S.pitch = S.tag * (1.0 + 0.001 * sp->fid_gap_permille);
  S.fhx = 0.5 * (sp->fid_grid_x * S.pitch - (S.pitch - S.tag)) + 0.5 * S.tag;   /* half extent incl. half-tag quiet zone */
  S.fhy = 0.5 * (sp->fid_grid_y * S.pitch - (S.pitch - S.tag)) + 0.5 * S.tag;
  S.nsx = sp->board_cols + 1; S.nsy = sp->board_rows + 1;   /* squares */
  S.hx = 0.5 * S.nsx; S.hy = 0.5 * S.nsy;                   /* half extents in squares */
  S.mg = (double)sp->margin_squares;
  const int ss = S.ss;
  const uint64_t key = splitmix64(sp->seed + (uint64_t)frame_index);
  const int nch = (cfg->pixfmt == RCC_PIX_MONO8) ? 1 : 3;
  const int rgb = (cfg->pixfmt == RCC_PIX_RGB8);       /* the three values of a pixel stored in the opposite order */
  int taps[RCC_SYNTH_BLUR_TAPS];
  const int optics = orc_synth_optics(sp, taps);
  if (optics < 0) return -1;
  if (!optics) {
    /* the ideal camera (ABI 1): mean of the samples + noise */
    for (int v = 0; v < h; ++v)
      for (int u = 0; u < w; ++u) {
        double acc[3];
        sample_pixel(&S, u, v, acc);
        size_t pix = (size_t)v * w + u;
        for (int c = 0; c < nch; ++c) {
          double val = acc[nch == 3 ? c : 1] / (double)(ss * ss);
          val += sp->noise_sigma * hash_gauss(key, (uint64_t)pix * 3u + (uint64_t)c);
          double rr = rint(val);
          int iv = rr < 0.0 ? 0 : (rr > 255.0 ? 255 : (int)rr);
          out[(size_t)v * cfg->stride_bytes + (size_t)u * nch + ((rgb && nch == 3) ? 2 - c : c)] = (uint8_t)iv;
        }
      }
    return 0;
  }
  /* optics on.  Pass 1: the supersampled sums as integers (negative sums -- a dark class colour with a negative tint -- are kept). */
  int32_t* A = (int32_t*)malloc((size_t)w * h * 3 * sizeof(int32_t));
  int32_t* T = (int32_t*)malloc((size_t)w * h * 3 * sizeof(int32_t));
  if (!A || !T) { free(A); free(T); return -2; }
  for (int v = 0; v < h; ++v)
    for (int u = 0; u < w; ++u) {
      double acc[3];
      sample_pixel(&S, u, v, acc);
      for (int c = 0; c < 3; ++c) A[((size_t)v * w + u) * 3 + c] = (int32_t)acc[c];
    }
  /* Pass 2: the separable filter along the rows (weights in 1/256, columns clamped at the border) ... */
  for (int v = 0; v < h; ++v)
    for (int u = 0; u < w; ++u) {
      int32_t r[3] = { 0, 0, 0 };
      for (int i = -(RCC_SYNTH_BLUR_TAPS - 1); i < RCC_SYNTH_BLUR_TAPS; ++i) {
        const int wi = taps[i < 0 ? -i : i];
        if (!wi) continue;
        const int32_t* a = &A[((size_t)v * w + clampi(u + i, 0, w - 1)) * 3];
        r[0] += wi * a[0]; r[1] += wi * a[1]; r[2] += wi * a[2];
      }
      int32_t* t = &T[((size_t)v * w + u) * 3];
      t[0] = r[0]; t[1] = r[1]; t[2] = r[2];
    }
  /* ... then along the columns (1/256 again, rows clamped), the gain in 1/4096, one division, noise, rounding */
  const double scale = 256.0 * 256.0 * 4096.0 * (double)(ss * ss);
  for (int v = 0; v < h; ++v)
    for (int u = 0; u < w; ++u) {
      int64_t b[3] = { 0, 0, 0 };
      for (int j = -(RCC_SYNTH_BLUR_TAPS - 1); j < RCC_SYNTH_BLUR_TAPS; ++j) {
        const int wj = taps[j < 0 ? -j : j];
        if (!wj) continue;
        const int32_t* t = &T[((size_t)clampi(v + j, 0, h - 1) * w + u) * 3];
        b[0] += (int64_t)wj * t[0]; b[1] += (int64_t)wj * t[1]; b[2] += (int64_t)wj * t[2];
      }
      const int64_t g = shading_gain(sp, w, h, u, v);
      size_t pix = (size_t)v * w + u;
      for (int c = 0; c < nch; ++c) {
        double val = (double)(b[nch == 3 ? c : 1] * g) / scale;
        val += sp->noise_sigma * hash_gauss(key, (uint64_t)pix * 3u + (uint64_t)c);
        double rr = rint(val);
        int iv = rr < 0.0 ? 0 : (rr > 255.0 ? 255 : (int)rr);
        out[(size_t)v * cfg->stride_bytes + (size_t)u * nch + ((rgb && nch == 3) ? 2 - c : c)] = (uint8_t)iv;
      }
    }
  free(A); free(T);
  return 0;
}
