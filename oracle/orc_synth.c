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

void orc_synth_render(const rcc_config* cfg, const rcc_synth_params* sp, const double pose[6],
                      int frame_index, uint8_t* out)
{
  const int w = cfg->width, h = cfg->height;
  const double fx = cfg->K[0], cx = cfg->K[2], fy = cfg->K[4], cy = cfg->K[5];
  double R[9];
  orc_rodrigues_v2m(pose, R, NULL);
  double H[9] = { R[0], R[1], pose[3], R[3], R[4], pose[4], R[6], R[7], pose[5] };
  double Hi[9];
  inv3(H, Hi);
  const int ss = sp->supersample < 1 ? 1 : sp->supersample;
  const double sq = sp->board_square;
  const int fid = sp->fid_grid_x > 0 && sp->fid_grid_y > 0;       /* planar grid of fiducials instead of the board */
  const double tag = cfg->tag_size, pitch = tag * (1.0 + 0.001 * sp->fid_gap_permille);
  const double fhx = 0.5 * (sp->fid_grid_x * pitch - (pitch - tag)) + 0.5 * tag;   /* half extent incl. half-tag quiet zone */
  const double fhy = 0.5 * (sp->fid_grid_y * pitch - (pitch - tag)) + 0.5 * tag;
  const int nsx = sp->board_cols + 1, nsy = sp->board_rows + 1;   /* squares */
  const double hx = 0.5 * nsx, hy = 0.5 * nsy;                   /* half extents in squares */
  const double mg = (double)sp->margin_squares;
  /* class colours (B,G,R): slight tints so the grey conversion is exercised */
  const int base[3] = { sp->background, sp->white, sp->black };
  const int tint[3][3] = { { 10, 0, -10 }, { -8, 0, -3 }, { 4, 0, 2 } };
  const uint64_t key = splitmix64(sp->seed + (uint64_t)frame_index);
  const int nch = (cfg->pixfmt == RCC_PIX_BGR8) ? 3 : 1;
  for (int v = 0; v < h; ++v)
    for (int u = 0; u < w; ++u) {
      double acc[3] = { 0, 0, 0 };
      for (int sy = 0; sy < ss; ++sy)
        for (int sx = 0; sx < ss; ++sx) {
          double us = (double)u + ((double)sx + 0.5) / ss - 0.5;
          double vs = (double)v + ((double)sy + 0.5) / ss - 0.5;
          double xd = (us - cx) / fx, yd = (vs - cy) / fy, x, y;
          int cls = 0;
          if (undistort_norm(cfg->dist_model, cfg->D, xd, yd, &x, &y)) {
            double q0 = Hi[0] * x + Hi[1] * y + Hi[2];
            double q1 = Hi[3] * x + Hi[4] * y + Hi[5];
            double q2 = Hi[6] * x + Hi[7] * y + Hi[8];
            if (q2 > 0.0 && fid) {
              /* metres on the tag plane, x right, y up (object frame of camera_pose.cpp:158-161) */
              const double Xm = q0 / q2, Ym = q1 / q2;
              if (fabs(Xm) < fhx && fabs(Ym) < fhy) {
                cls = 1;
                const double gx = (Xm + fhx - 0.5 * tag) / pitch, gy = (fhy - 0.5 * tag - Ym) / pitch;   /* grid coords, row 0 on top */
                const int ti = (int)floor(gx), tj = (int)floor(gy);
                if (ti >= 0 && tj >= 0 && ti < sp->fid_grid_x && tj < sp->fid_grid_y) {
                  const double u = (gx - ti) * pitch / tag * 8.0, v = (gy - tj) * pitch / tag * 8.0;   /* cells */
                  if (u < 8.0 && v < 8.0) {
                    const int cu = (int)floor(u), cv = (int)floor(v);
                    if (cu == 0 || cv == 0 || cu == 7 || cv == 7) cls = 2;
                    else {
                      const uint64_t code = cfg->family_codes[(tj * sp->fid_grid_x + ti) % cfg->family_n];
                      cls = ((code >> (35 - ((cv - 1) * 6 + (cu - 1)))) & 1u) ? 1 : 2;
                    }
                  }
                }
              }
            } else if (q2 > 0.0) {
              double X = q0 / q2 / sq, Y = q1 / q2 / sq;   /* in squares, origin at the centre */
              if (fabs(X) < hx + mg && fabs(Y) < hy + mg) {
                cls = 1;
                if (fabs(X) < hx && fabs(Y) < hy) {
                  int i = (int)floor(X + hx), j = (int)floor(Y + hy);
                  cls = ((i + j) & 1) ? 1 : 2;
                }
              }
            }
          }
          for (int c = 0; c < 3; ++c) acc[c] += (double)(base[cls] + tint[cls][c]);
        }
      size_t pix = (size_t)v * w + u;
      for (int c = 0; c < nch; ++c) {
        double val = acc[nch == 3 ? c : 1] / (double)(ss * ss);
        val += sp->noise_sigma * hash_gauss(key, (uint64_t)pix * 3u + (uint64_t)c);
        double rr = rint(val);
        int iv = rr < 0.0 ? 0 : (rr > 255.0 ? 255 : (int)rr);
        out[(size_t)v * cfg->stride_bytes + (size_t)u * nch + c] = (uint8_t)iv;
      }
    }
}
