/*
 * orc_image.c -- CPU oracle, image stages a1..a4.  TEST INFRASTRUCTURE.
 *
 * None of this arithmetic exists in /root/reference: the reference subscribes to the output of an
 * external detector (real_preprocessing/src/corner_detections.cpp:4,41,78) that it launches on
 * image_raw (real_preprocessing/README.md:65).  The stages follow SURVEY.md appendix B, the
 * restated published definitions; where the build had to choose, the choice is marked [B] and
 * repeated in DESIGN.md.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <limits.h>
#include "orc.h"

/* ------------------------------------------------------------------------------------------------
 * a1  BGR8 -> grey.  SURVEY appendix B.1: Y = (B*1868 + G*9617 + R*4899 + (1<<13)) >> 14
 * (the image the detector receives is bgr8 from cv_camera, README.md:64-65).
 * ---------------------------------------------------------------------------------------------- */
void orc_bgr_to_grey(const uint8_t* bgr, int w, int h, int stride, uint8_t* grey)
{
  for (int y = 0; y < h; ++y) {
    const uint8_t* row = bgr + (size_t)y * stride;
    for (int x = 0; x < w; ++x) {
      int b = row[3 * x], g = row[3 * x + 1], r = row[3 * x + 2];
      grey[(size_t)y * w + x] = (uint8_t)((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14);
    }
  }
}
/* RCC_PIX_RGB8 (sensor_msgs "rgb8"): byte 0 is red */
void orc_rgb_to_grey(const uint8_t* rgb, int w, int h, int stride, uint8_t* grey)
{
  for (int y = 0; y < h; ++y) {
    const uint8_t* row = rgb + (size_t)y * stride;
    for (int x = 0; x < w; ++x) {
      int r = row[3 * x], g = row[3 * x + 1], b = row[3 * x + 2];
      grey[(size_t)y * w + x] = (uint8_t)((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14);
    }
  }
}

/* ------------------------------------------------------------------------------------------------
 * atan for r >= 0 from +,-,*,/ only, so that a device restatement is bit-identical [B].
 * Reduction: r>1 -> 1/r ; t>tan(pi/8) -> (t-1)/(t+1) ; odd Taylor series, 24 terms, Horner.
 * ---------------------------------------------------------------------------------------------- */
double orc_atan_pos(double r)
{
  const double PI_2 = 1.57079632679489661923, PI_4 = 0.78539816339744830962;
  const double T8 = 0.41421356237309504880;
  int flip = 0;
  double t = r;
  if (t > 1.0) { t = 1.0 / t; flip = 1; }
  double base = 0.0;
  if (t > T8) { t = (t - 1.0) / (t + 1.0); base = PI_4; }
  double z = t * t;
  double s = 0.0;
  for (int k = 23; k >= 0; --k) {
    double c = 1.0 / (double)(2 * k + 1);
    if (k & 1) c = -c;
    s = s * z + c;
  }
  double a = base + t * s;
  if (flip) a = PI_2 - a;
  return a;
}

static int32_t sat_rint_i32(double v)
{
  double r = rint(v);
  if (!(r > -2147483648.0)) return INT32_MIN;  /* also catches NaN */
  if (r > 2147483647.0) return INT32_MAX;
  return (int32_t)r;
}

/* ------------------------------------------------------------------------------------------------
 * a2  undistortion map, Q5 fixed point.  SURVEY appendix B.2 (initUndistortRectifyMap with
 * newK = K, R = I).  K and D are laid out as camera_pose.cpp:59-64 loads them: K row-major 3x3,
 * D = (k1,k2,p1,p2,k3).  Fisheye D = (k1..k4) is an extension [B]; the reference has none.
 * Every line is one rounded IEEE operation (or a short chain written in evaluation order).
 * ---------------------------------------------------------------------------------------------- */
void orc_undistort_map_q5(const double K[9], int dist_model, const double D[8], int w, int h,
                          int32_t* mapx, int32_t* mapy)
{
  const double fx = K[0], cx = K[2], fy = K[4], cy = K[5];
  for (int v = 0; v < h; ++v) {
    for (int u = 0; u < w; ++u) {
      double x = ((double)u - cx) / fx;
      double y = ((double)v - cy) / fy;
      double xs, ys;
      if (dist_model == RCC_DIST_PLUMB_BOB) {
        const double k1 = D[0], k2 = D[1], p1 = D[2], p2 = D[3], k3 = D[4];
        double x2 = x * x, y2 = y * y;
        double r2 = x2 + y2;
        double _2xy = (2.0 * x) * y;
        double kr = k3 * r2;
        kr = kr + k2;
        kr = kr * r2;
        kr = kr + k1;
        kr = kr * r2;
        kr = 1.0 + kr;
        double tx = 2.0 * x2;
        tx = r2 + tx;
        double ty = 2.0 * y2;
        ty = r2 + ty;
        double xd = x * kr;
        double a = p1 * _2xy;
        xd = xd + a;
        a = p2 * tx;
        xd = xd + a;
        double yd = y * kr;
        a = p1 * ty;
        yd = yd + a;
        a = p2 * _2xy;
        yd = yd + a;
        xs = fx * xd;
        xs = xs + cx;
        ys = fy * yd;
        ys = ys + cy;
      } else if (dist_model == RCC_DIST_FISHEYE) {
        const double k1 = D[0], k2 = D[1], k3 = D[2], k4 = D[3];
        double x2 = x * x, y2 = y * y;
        double r = sqrt(x2 + y2);
        double th = orc_atan_pos(r);
        double t2 = th * th;
        double p = k4 * t2;
        p = p + k3;
        p = p * t2;
        p = p + k2;
        p = p * t2;
        p = p + k1;
        p = p * t2;
        p = 1.0 + p;
        double thd = th * p;
        double s = (r > 1e-8) ? thd / r : 1.0;
        xs = fx * x;
        xs = xs * s;
        xs = xs + cx;
        ys = fy * y;
        ys = ys * s;
        ys = ys + cy;
      } else {
        xs = (double)u;
        ys = (double)v;
      }
      mapx[(size_t)v * w + u] = sat_rint_i32(xs * 32.0);
      mapy[(size_t)v * w + u] = sat_rint_i32(ys * 32.0);
    }
  }
}

/* remap, INTER_LINEAR with 5 fractional bits, BORDER_CONSTANT 0 (appendix B.2).  With 5-bit
 * fractions the Q15 weight table is exactly 32*(32-fx)(32-fy) etc. and sums to 32768, so
 * (sum w*p + (1<<14)) >> 15 == (sum wq*p + 512) >> 10 with wq the 10-bit products. */
void orc_remap_q5(const uint8_t* src, int w, int h, int stride, const int32_t* mapx,
                  const int32_t* mapy, uint8_t* dst)
{
  for (int v = 0; v < h; ++v) {
    for (int u = 0; u < w; ++u) {
      int32_t X = mapx[(size_t)v * w + u], Y = mapy[(size_t)v * w + u];
      int32_t ix = X >> 5, iy = Y >> 5;
      int32_t fx = X & 31, fy = Y & 31;
      int p00 = 0, p01 = 0, p10 = 0, p11 = 0;
      if (iy >= 0 && iy < h) {
        if (ix >= 0 && ix < w) p00 = src[(size_t)iy * stride + ix];
        if (ix + 1 >= 0 && ix + 1 < w) p01 = src[(size_t)iy * stride + ix + 1];
      }
      if (iy + 1 >= 0 && iy + 1 < h) {
        if (ix >= 0 && ix < w) p10 = src[(size_t)(iy + 1) * stride + ix];
        if (ix + 1 >= 0 && ix + 1 < w) p11 = src[(size_t)(iy + 1) * stride + ix + 1];
      }
      int acc = (32 - fx) * (32 - fy) * p00 + fx * (32 - fy) * p01 + (32 - fx) * fy * p10 +
                fx * fy * p11;
      dst[(size_t)v * w + u] = (uint8_t)((acc + 512) >> 10);
    }
  }
}

/* a1+a2 for one frame: grey first, then remap of the grey image [B] (the order is the build's
 * choice: the reference never undistorts pixels). */
int orc_ingest(const rcc_config* cfg, const uint8_t* frame, uint8_t* grey_out)
{
  const int w = cfg->width, h = cfg->height;
  uint8_t* grey = grey_out;
  uint8_t* tmp = NULL;
  const int undist = cfg->undistort && cfg->dist_model != RCC_DIST_NONE;
  if (undist) {
    tmp = (uint8_t*)malloc((size_t)w * h);
    if (!tmp) return RCC_ERR_NOMEM;
    grey = tmp;
  }
  if (cfg->pixfmt == RCC_PIX_BGR8) {
    orc_bgr_to_grey(frame, w, h, cfg->stride_bytes, grey);
  } else if (cfg->pixfmt == RCC_PIX_RGB8) {
    orc_rgb_to_grey(frame, w, h, cfg->stride_bytes, grey);
  } else {
    for (int y = 0; y < h; ++y) memcpy(grey + (size_t)y * w, frame + (size_t)y * cfg->stride_bytes, w);
  }
  if (undist) {
    int32_t* mx = (int32_t*)malloc(sizeof(int32_t) * (size_t)w * h);
    int32_t* my = (int32_t*)malloc(sizeof(int32_t) * (size_t)w * h);
    if (!mx || !my) { free(mx); free(my); free(tmp); return RCC_ERR_NOMEM; }
    orc_undistort_map_q5(cfg->K, cfg->dist_model, cfg->D, w, h, mx, my);
    orc_remap_q5(tmp, w, h, w, mx, my, grey_out);
    free(mx); free(my); free(tmp);
  }
  return RCC_OK;
}

/* ------------------------------------------------------------------------------------------------
 * a3  apriltag tile threshold, SURVEY appendix B.3.  Tiles are 4x4; each tile takes the max of
 * max / min of min over its 3x3 tile neighbourhood (clamped at the borders); per pixel:
 * max-min < min_contrast -> 127, else v > min + (max-min)/2 ? 255 : 0.
 * [B] ragged sizes: the last tile of a row/column is extended to the image edge (its extra
 * pixels contribute to its min/max), so every pixel belongs to exactly one tile.
 * ---------------------------------------------------------------------------------------------- */
void orc_threshold_tiles(const uint8_t* grey, int w, int h, int min_contrast, uint8_t* bin)
{
  int tw = w / 4, th = h / 4;
  if (tw < 1) tw = 1;
  if (th < 1) th = 1;
  uint8_t* tmin = (uint8_t*)malloc((size_t)tw * th);
  uint8_t* tmax = (uint8_t*)malloc((size_t)tw * th);
  uint8_t* dmin = (uint8_t*)malloc((size_t)tw * th);
  uint8_t* dmax = (uint8_t*)malloc((size_t)tw * th);
  memset(tmin, 255, (size_t)tw * th);
  memset(tmax, 0, (size_t)tw * th);
  for (int y = 0; y < h; ++y) {
    int ty = y >> 2; if (ty > th - 1) ty = th - 1;
    for (int x = 0; x < w; ++x) {
      int tx = x >> 2; if (tx > tw - 1) tx = tw - 1;
      uint8_t v = grey[(size_t)y * w + x];
      size_t t = (size_t)ty * tw + tx;
      if (v < tmin[t]) tmin[t] = v;
      if (v > tmax[t]) tmax[t] = v;
    }
  }
  for (int ty = 0; ty < th; ++ty)
    for (int tx = 0; tx < tw; ++tx) {
      int mn = 255, mx = 0;
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          int yy = ty + dy, xx = tx + dx;
          if (yy < 0 || yy >= th || xx < 0 || xx >= tw) continue;
          int a = tmin[(size_t)yy * tw + xx], b = tmax[(size_t)yy * tw + xx];
          if (a < mn) mn = a;
          if (b > mx) mx = b;
        }
      dmin[(size_t)ty * tw + tx] = (uint8_t)mn;
      dmax[(size_t)ty * tw + tx] = (uint8_t)mx;
    }
  for (int y = 0; y < h; ++y) {
    int ty = y >> 2; if (ty > th - 1) ty = th - 1;
    for (int x = 0; x < w; ++x) {
      int tx = x >> 2; if (tx > tw - 1) tx = tw - 1;
      int mn = dmin[(size_t)ty * tw + tx], mx = dmax[(size_t)ty * tw + tx];
      int v = grey[(size_t)y * w + x];
      uint8_t o;
      if (mx - mn < min_contrast) o = 127;
      else o = (v > mn + (mx - mn) / 2) ? 255 : 0;
      bin[(size_t)y * w + x] = o;
    }
  }
  free(tmin); free(tmax); free(dmin); free(dmax);
}

/* ------------------------------------------------------------------------------------------------
 * a4  Harris response, all-integer form [B] of SURVEY appendix B.4 (cornerHarris structure:
 * Sobel 3x3 -> structure tensor over blockSize 5 -> det - k*trace^2), evaluated on the EVEN pixel
 * lattice (x and y even):
 *   gx, gy = Sobel3x3 >> 3          (arithmetic shift; grey levels per pixel, in [-128,127])
 *   A = (sum_5x5 gx*gx) >> 4,  B = (sum_5x5 gx*gy) >> 4 (floor),  C = (sum_5x5 gy*gy) >> 4
 *                                                                      (0 <= A,C <= 25600)
 *   R = A*C - B*B - ((A+C)^2 >> 4)                                    (k = 1/16), fits int32
 * R is defined at even (x,y) whose 7x7 support lies inside the image; elsewhere INT32_MIN.
 * The lattice halves the response work per axis; sub-pixel refinement (a5) recovers the position.
 * ---------------------------------------------------------------------------------------------- */
void orc_harris_response(const uint8_t* g, int w, int h, int32_t* R)
{
  size_t n = (size_t)w * h;
  int8_t* gxv = (int8_t*)calloc(n, 1);
  int8_t* gyv = (int8_t*)calloc(n, 1);
  for (size_t i = 0; i < n; ++i) R[i] = INT32_MIN;
  for (int y = 1; y < h - 1; ++y)
    for (int x = 1; x < w - 1; ++x) {
      const uint8_t* p = g + (size_t)y * w + x;
      int sx = (p[-w + 1] + 2 * p[1] + p[w + 1]) - (p[-w - 1] + 2 * p[-1] + p[w - 1]);
      int sy = (p[w - 1] + 2 * p[w] + p[w + 1]) - (p[-w - 1] + 2 * p[-w] + p[-w + 1]);
      gxv[(size_t)y * w + x] = (int8_t)(sx >> 3);
      gyv[(size_t)y * w + x] = (int8_t)(sy >> 3);
    }
  for (int y = 4; y < h - 3; y += 2)
    for (int x = 4; x < w - 3; x += 2) {
      int32_t sa = 0, sb = 0, sc = 0;
      for (int dy = -2; dy <= 2; ++dy)
        for (int dx = -2; dx <= 2; ++dx) {
          size_t i = (size_t)(y + dy) * w + (x + dx);
          int a = gxv[i], b = gyv[i];
          sa += a * a; sb += a * b; sc += b * b;
        }
      int32_t A = sa >> 4, B = sb >> 4, C = sc >> 4;
      uint32_t tr = (uint32_t)(A + C);
      R[(size_t)y * w + x] = A * C - B * B - (int32_t)((tr * tr) >> 4);
    }
  free(gxv); free(gyv);
}

/* a4 dense selection [B]: lattice points with R >= thresh that are 3x3 maxima ON THE LATTICE
 * (neighbours at +-2 pixels) with scan-order tie break: greater than the four lattice neighbours
 * that precede it in (y,x) order, not less than the four that follow; at least `margin` pixels
 * from every border.  Output sorted by (y,x). */
int orc_harris_candidates(const int32_t* R, int w, int h, int thresh, int margin, orc_cand* out, int cap)
{
  int n = 0;
  if (margin < 6) margin = 6;
  int m0 = (margin + 1) & ~1;   /* first even coordinate >= margin */
  for (int y = m0; y < h - margin; y += 2)
    for (int x = m0; x < w - margin; x += 2) {
      const int32_t* p = R + (size_t)y * w + x;
      const int w2 = 2 * w;
      int32_t r = *p;
      if (r < thresh) continue;
      if (!(r > p[-w2 - 2] && r > p[-w2] && r > p[-w2 + 2] && r > p[-2])) continue;
      if (!(r >= p[2] && r >= p[w2 - 2] && r >= p[w2] && r >= p[w2 + 2])) continue;
      if (n < cap) { out[n].x = (int16_t)x; out[n].y = (int16_t)y; out[n].score = r; }
      ++n;
    }
  return n;
}

/* X-junction ring test [B] (appendix B.4 last sentence): 16 samples of the threshold map on a
 * radius-5 ring; a checkerboard inner corner shows exactly 4 black/white transitions and no
 * low-contrast (127) sample. */
static const int8_t RING16[16][2] = {
  { 5, 0}, { 5, 2}, { 4, 4}, { 2, 5}, { 0, 5}, {-2, 5}, {-4, 4}, {-5, 2},
  {-5, 0}, {-5,-2}, {-4,-4}, {-2,-5}, { 0,-5}, { 2,-5}, { 4,-4}, { 5,-2}
};

int orc_xjunction_ring(const uint8_t* bin, int w, int h, int x, int y)
{
  if (x < 5 || y < 5 || x >= w - 5 || y >= h - 5) return 0;
  int v[16];
  for (int k = 0; k < 16; ++k) {
    v[k] = bin[(size_t)(y + RING16[k][1]) * w + (x + RING16[k][0])];
    if (v[k] == 127) return 0;
  }
  int tr = 0;
  for (int k = 0; k < 16; ++k) tr += (v[k] != v[(k + 1) & 15]);
  return tr == 4;
}

/* The same junction read off the GREY image against the ring's own mid level [B] (round 4).  The threshold map decides "flat"
 * per 4x4 tile from a 12x12 neighbourhood: with a low min_contrast, sensor noise in a plain area turns into salt and pepper, and
 * a ring that crosses such an area can show four transitions by accident (seen on the L-shaped outer corners of the board, which
 * sit ON the lattice of inner corners: one such point makes the lattice stage count 49).  Against the ring's own mid level
 * ((min + max) >> 1) noise in a plain area is far from the level: the ring must span min_contrast and show exactly four
 * transitions too.  a4.3 asks for BOTH tests. */
int orc_xjunction_ring_grey(const uint8_t* grey, int w, int h, int x, int y, int min_contrast)
{
  if (x < 5 || y < 5 || x >= w - 5 || y >= h - 5) return 0;
  int g[16], lo = 255, hi = 0;
  for (int k = 0; k < 16; ++k) {
    g[k] = grey[(size_t)(y + RING16[k][1]) * w + (x + RING16[k][0])];
    if (g[k] < lo) lo = g[k];
    if (g[k] > hi) hi = g[k];
  }
  if (hi - lo < min_contrast) return 0;
  const int mid = (lo + hi) >> 1;
  int tr = 0;
  for (int k = 0; k < 16; ++k) tr += ((g[k] > mid) != (g[(k + 1) & 15] > mid));
  return tr == 4;
}

/* a5's gate for board scenes [B] (round 4): is the candidate worth refining?  Besides its 48 inner corners a 9 x 7-square board
 * raises 36 Harris candidates on its outline -- L-shaped corners of single squares against the white margin -- which a5 refined at
 * the cost of the true ones (4-5 iterations each: 42 % of the stage's work) only for a4.3 to reject them.  A radius-11 ring around the
 * UNREFINED pixel (a Harris maximum sits up to 3 px off its junction on a sharp image, up to 6.5 px under a Gaussian blur of sigma
 * 2.3: the ring still encloses it) read against its own mid level shows four or more transitions at a junction and two at an
 * L-corner or an edge.  Candidates that show fewer than four are not refined and reach a4.3 as (-1, -1), which it rejects.  A ring
 * that leaves the image passes (a5 and a4.3 decide).  Measured over the 12 288 corners of 256 rendered 1080p frames
 * (scratch/gate_study.py): radius 8 / 9 / 10 / 11 hold back 52 / 17 / 0 / 0 inner corners at sigma 2.0 and 139 / 50 / 1 / 0 at
 * sigma 2.3; every outline candidate is held back at any of them. */
static const int8_t RING11[16][2] = {
  {11, 0}, {10, 4}, { 8, 8}, { 4,10}, { 0,11}, {-4,10}, {-8, 8}, {-10, 4},
  {-11, 0}, {-10,-4}, {-8,-8}, {-4,-10}, { 0,-11}, { 4,-10}, { 8,-8}, {10,-4}
};
int orc_junction_pretest(const uint8_t* grey, int w, int h, int x, int y, int min_contrast)
{
  if (x < 11 || y < 11 || x >= w - 11 || y >= h - 11) return 1;
  int g[16], lo = 255, hi = 0;
  for (int k = 0; k < 16; ++k) {
    g[k] = grey[(size_t)(y + RING11[k][1]) * w + (x + RING11[k][0])];
    if (g[k] < lo) lo = g[k];
    if (g[k] > hi) hi = g[k];
  }
  if (hi - lo < min_contrast) return 0;
  const int mid = (lo + hi) >> 1;
  int tr = 0;
  for (int k = 0; k < 16; ++k) tr += ((g[k] > mid) != (g[(k + 1) & 15] > mid));
  return tr >= 4;
}

/* list-level suppression [B]: entry i survives iff no other entry j within Chebyshev distance
 * nms_radius has a larger score (or an equal score and a smaller index).  Not greedy: decided
 * against the full input list, so it is order-independent apart from the tie break.  Survivors
 * are then ring-validated.  Input and output sorted by (y,x). */
int orc_filter_candidates(const orc_cand* in, int n, const uint8_t* bin, int w, int h,
                          int nms_radius, int xj_check, orc_cand* out, int cap)
{
  int m = 0;
  for (int i = 0; i < n; ++i) {
    int keep = 1;
    for (int j = 0; j < n && keep; ++j) {
      if (j == i) continue;
      int dx = in[j].x - in[i].x, dy = in[j].y - in[i].y;
      if (dx < 0) dx = -dx;
      if (dy < 0) dy = -dy;
      if (dx <= nms_radius && dy <= nms_radius) {
        if (in[j].score > in[i].score || (in[j].score == in[i].score && j < i)) keep = 0;
      }
    }
    if (keep && xj_check) keep = orc_xjunction_ring(bin, w, h, in[i].x, in[i].y);
    if (keep) {
      if (m < cap) out[m] = in[i];
      ++m;
    }
  }
  return m;
}
