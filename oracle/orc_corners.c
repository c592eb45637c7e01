/*
 * orc_corners.c -- CPU oracle, stages a5 (sub-pixel refinement) and a6 (board indexing).
 * TEST INFRASTRUCTURE.  The consumer side of these outputs in the reference:
 * real_preprocessing/src/corner_detections.cpp:48-55 (id, four pixel corners, cast to int) and
 * real_preprocessing/src/camera_pose.cpp:152-161 (corner order bl,br,tr,tl; object frame x right,
 * y up, z = 0, origin at the centre).
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>
#include "orc.h"

/* ------------------------------------------------------------------------------------------------
 * a5  cornerSubPix form, SURVEY appendix B.5, with zeroZone = (-1,-1).
 * [B] numerics: everything in binary64; the five window sums are accumulated in 64 bins
 * (bin = sample index mod 64, samples in increasing index order) and the bins are combined by
 * a balanced pairwise tree  v[l] += v[l ^ off], off = 32,16,...,1 .  Every operation is a single
 * rounded IEEE operation, so a 64-lane wavefront doing one bin per lane and an xor-butterfly
 * reproduces the result bit for bit.
 * The Gaussian-like mask is exp(-(i/w)^2) * exp(-(j/w)^2), the 1-D factors computed once with the
 * host libm (the product library does the same on the host and uploads the table).
 * ---------------------------------------------------------------------------------------------- */
#define SP_MAXW 7
#define GMAXPTS_V 2048        /* entries of the suppressed list a4.3 looks at (RCC_MAX_KEPT_FIDUCIAL) */

static double tree64(double* v)
{
  for (int off = 32; off >= 1; off >>= 1) {
    double t[64];
    for (int l = 0; l < 64; ++l) t[l] = v[l] + v[l ^ off];
    memcpy(v, t, sizeof(t));
  }
  return v[0];
}

void orc_subpix_mask(int win, double* m1 /* 2*win+1 */)
{
  for (int k = -win; k <= win; ++k) {
    double t = (double)k / (double)win;
    m1[k + win] = exp(-(t * t));
  }
}

void orc_corner_subpix(const uint8_t* g, int w, int h, const orc_cand* pts, int n, int win,
                       int max_iter, double eps, double* xy_out)
{
  if (win < 1) win = 1;
  if (win > SP_MAXW) win = SP_MAXW;
  const int ww = 2 * win + 1;     /* window side */
  const int pw = 2 * win + 3;     /* interpolated patch side */
  double m1[2 * SP_MAXW + 1];
  orc_subpix_mask(win, m1);
  double S[(2 * SP_MAXW + 3) * (2 * SP_MAXW + 3)];
  const double eps2 = eps * eps;
  for (int q = 0; q < n; ++q) {
    const double x0 = (double)pts[q].x, y0 = (double)pts[q].y;
    double cx = x0, cy = y0;
    int iter = 0;
    int bad = 0;
    double err;
    do {
      double flx = floor(cx), fly = floor(cy);
      int ix = (int)flx, iy = (int)fly;
      if (ix - win - 1 < 0 || iy - win - 1 < 0 || ix + win + 2 > w - 1 || iy + win + 2 > h - 1) {
        bad = 1;
        break;
      }
      double fx = cx - flx, fy = cy - fly;
      double ofx = 1.0 - fx, ofy = 1.0 - fy;
      double a00 = ofx * ofy, a01 = fx * ofy, a10 = ofx * fy, a11 = fx * fy;
      for (int i = 0; i < pw; ++i)
        for (int j = 0; j < pw; ++j) {
          const uint8_t* p = g + (size_t)(iy + i - win - 1) * w + (ix + j - win - 1);
          double t0 = a00 * (double)p[0];
          double t1 = a01 * (double)p[1];
          double t2 = a10 * (double)p[w];
          double t3 = a11 * (double)p[w + 1];
          double s = t0 + t1;
          s = s + t2;
          s = s + t3;
          S[i * pw + j] = s;
        }
      double ba[64], bb[64], bc[64], b1[64], b2[64];
      for (int l = 0; l < 64; ++l) ba[l] = bb[l] = bc[l] = b1[l] = b2[l] = 0.0;
      for (int k = 0; k < ww * ww; ++k) {
        int i = k / ww, j = k % ww;           /* window row, column, 0-based */
        const double* sp = S + (i + 1) * pw + (j + 1);
        double gx = sp[1] - sp[-1];
        double gy = sp[pw] - sp[-pw];
        double m = m1[i] * m1[j];
        double gxx = (gx * gx) * m;
        double gxy = (gx * gy) * m;
        double gyy = (gy * gy) * m;
        double px = (double)(j - win), py = (double)(i - win);
        int l = k & 63;
        ba[l] = ba[l] + gxx;
        bb[l] = bb[l] + gxy;
        bc[l] = bc[l] + gyy;
        double u1 = gxx * px, u2 = gxy * py;
        b1[l] = b1[l] + (u1 + u2);
        double v1 = gxy * px, v2 = gyy * py;
        b2[l] = b2[l] + (v1 + v2);
      }
      double a = tree64(ba), b = tree64(bb), c = tree64(bc), bb1 = tree64(b1), bb2 = tree64(b2);
      double ac = a * c, b2_ = b * b;
      double det = ac - b2_;
      if (fabs(det) <= DBL_EPSILON * DBL_EPSILON) break;
      double scale = 1.0 / det;
      double cs = c * scale, bs = b * scale, as = a * scale;
      double dx = cs * bb1 - bs * bb2;   /* two products, one subtraction, each rounded */
      double dy = as * bb2 - bs * bb1;
      double nx = cx + dx, ny = cy + dy;
      double ex = nx - cx, ey = ny - cy;
      err = ex * ex + ey * ey;
      cx = nx; cy = ny;
      if (cx < 0.0 || cx >= (double)w || cy < 0.0 || cy >= (double)h) break;
    } while (++iter < max_iter && err > eps2);
    if (bad || fabs(cx - x0) > (double)win || fabs(cy - y0) > (double)win) { cx = x0; cy = y0; }
    xy_out[2 * q] = cx;
    xy_out[2 * q + 1] = cy;
  }
}

/* ------------------------------------------------------------------------------------------------
 * a4.3 validation of refined corners [B].  Harris maxima sit up to ~3 px off an X-junction (the
 * gradient vanishes at its centre) and one junction can raise two maxima, so validation runs on
 * the REFINED position rounded to the nearest pixel:
 *   - X-junction ring tests there: on the threshold map (orc_xjunction_ring) AND on the grey image against the ring's own
 *     mid level (orc_xjunction_ring_grey),
 *   - de-duplication: drop i if another validated j lies within Chebyshev distance dedupe_radius
 *     of it and has a larger score (or an equal score and a smaller index).  Not greedy.
 * Output keeps the input order; out[k].x/.y are the rounded refined pixel, out_xy the refined
 * position.  Returns the number kept, or a number > cap when more than cap entries pass the ring tests (nothing usable is
 * written then).
 * ---------------------------------------------------------------------------------------------- */
int orc_validate_refined(const orc_cand* pre, int n, const double* xy, const uint8_t* bin, const uint8_t* grey, int w,
                         int h, int xj_check, int min_contrast, int dedupe_radius, orc_cand* out, double* out_xy, int cap)
{
  int16_t rx[GMAXPTS_V], ry[GMAXPTS_V];
  uint8_t ok[GMAXPTS_V];
  if (n > GMAXPTS_V) n = GMAXPTS_V;
  for (int i = 0; i < n; ++i) {
    int x = (int)floor(xy[2 * i] + 0.5), y = (int)floor(xy[2 * i + 1] + 0.5);
    rx[i] = (int16_t)x; ry[i] = (int16_t)y;
    int v = (x >= 5 && y >= 5 && x < w - 5 && y < h - 5);
    if (v && xj_check) v = orc_xjunction_ring(bin, w, h, x, y) && orc_xjunction_ring_grey(grey, w, h, x, y, min_contrast);
    ok[i] = (uint8_t)v;
  }
  /* capacity [B]: at most `cap` entries may pass the ring tests (what the lattice stage is built for); more -- a cluttered scene full
   * of junction-like points -- and the frame is rejected by the caller (the count returned exceeds cap) */
  {
    int nvalid = 0;
    for (int i = 0; i < n; ++i) nvalid += ok[i];
    if (nvalid > cap) return nvalid;
  }
  int m = 0;
  for (int i = 0; i < n; ++i) {
    if (!ok[i]) continue;
    int keep = 1;
    for (int j = 0; j < n && keep; ++j) {
      if (j == i || !ok[j]) continue;
      int dx = rx[j] - rx[i], dy = ry[j] - ry[i];
      if (dx < 0) dx = -dx;
      if (dy < 0) dy = -dy;
      if (dx <= dedupe_radius && dy <= dedupe_radius) {
        if (pre[j].score > pre[i].score || (pre[j].score == pre[i].score && j < i)) keep = 0;
      }
    }
    if (keep) {
      if (m < cap) {
        out[m].x = rx[i]; out[m].y = ry[i]; out[m].score = pre[i].score;
        out_xy[2 * m] = xy[2 * i]; out_xy[2 * m + 1] = xy[2 * i + 1];
      }
      ++m;
    }
  }
  return m;
}

/* ------------------------------------------------------------------------------------------------
 * a6  board indexing [B]: label the cols x rows lattice of inner corners by seeded growth on the
 * INTEGER candidate positions (so every decision is exact), then fix the orientation:
 *   - column axis x row axis has a positive cross product in image coordinates (y down), i.e.
 *     the board is seen from the front, columns to the right, rows downwards;
 *   - of the two remaining labelings (180 degrees apart -- a board with an odd x odd count of
 *     squares cannot tell them apart) keep the one whose corner 0 precedes its last corner in
 *     (y,x) order.
 * Output index = row*cols + col.  Object point of (col,row): ((col-(cols-1)/2)*sq,
 * ((rows-1)/2-row)*sq, 0): x right, y up, origin at the centre, as camera_pose.cpp:158-161.
 * ---------------------------------------------------------------------------------------------- */
#define GM 24                    /* labels span [-GM, GM] */
#define GBOARD 16                /* max inner corners per side */
#define GEXTRA 8    /* labels beyond cols x rows that the window rule of round 4 takes */
#define GW (2 * GM + 1)
#define GMAXPTS 256

typedef struct { int64_t x, y; } v2;

static inline int64_t d2(v2 a, v2 b) { int64_t dx = a.x - b.x, dy = a.y - b.y; return dx * dx + dy * dy; }

/* nearest point to `pred` among those with used[k]==0; tie -> lower index; -1 if none */
static int nearest_free(const v2* p, int n, const uint8_t* used, v2 pred, int64_t* dist)
{
  int best = -1;
  int64_t bd = 0;
  for (int k = 0; k < n; ++k) {
    if (used[k]) continue;
    int64_t d = d2(p[k], pred);
    if (best < 0 || d < bd) { best = k; bd = d; }
  }
  *dist = bd;
  return best;
}

int orc_grid_index(const orc_cand* pts, int n, int cols, int rows, int32_t* order_out)
{
  const int need = cols * rows;
  if (n < need || n > GMAXPTS || cols < 2 || rows < 2 || cols > GBOARD || rows > GBOARD) return 0;
  v2 p[GMAXPTS];
  int64_t sx = 0, sy = 0;
  for (int i = 0; i < n; ++i) { p[i].x = pts[i].x; p[i].y = pts[i].y; sx += p[i].x; sy += p[i].y; }

  /* seeds: the 8 points closest to the centroid, closest first (compare |n*p - sum|^2) */
  int seeds[16];
  int nseeds = 0;
  {
    uint8_t taken[GMAXPTS];
    memset(taken, 0, sizeof(taken));
    for (int s = 0; s < 8 && s < n; ++s) {
      int best = -1;
      int64_t bd = 0;
      for (int i = 0; i < n; ++i) {
        if (taken[i]) continue;
        int64_t ex = (int64_t)n * p[i].x - sx, ey = (int64_t)n * p[i].y - sy;
        int64_t d = ex * ex + ey * ey;
        if (best < 0 || d < bd) { best = i; bd = d; }
      }
      taken[best] = 1;
      seeds[nseeds++] = best;
    }
  }

  /* [B] round 4: should none of them grow the board -- clutter all over the scene pulls the centroid off it -- up to 8 more seeds
   * follow: the points with the largest score (Harris response) that have not been tried, largest first, ties to the smaller
   * index.  A board's corners are the strongest junctions of most scenes.  A frame whose board grows from a centroid seed is
   * untouched by this. */
  {
    uint8_t tried[GMAXPTS];
    memset(tried, 0, sizeof(tried));
    for (int q = 0; q < nseeds; ++q) tried[seeds[q]] = 1;
    const int ncentroid = nseeds;
    for (int s2 = 0; s2 < 8 && ncentroid + s2 < n; ++s2) {
      int best = -1;
      for (int i = 0; i < n; ++i) {
        if (tried[i]) continue;
        if (best < 0 || pts[i].score > pts[best].score) best = i;
      }
      tried[best] = 1;
      seeds[nseeds++] = best;
    }
  }

  for (int si = 0; si < nseeds; ++si) {
    const int s = seeds[si];
    uint8_t used[GMAXPTS];
    memset(used, 0, sizeof(used));
    used[s] = 1;
    int64_t dd;
    int n1 = nearest_free(p, n, used, p[s], &dd);
    if (n1 < 0) continue;
    v2 u = { p[n1].x - p[s].x, p[n1].y - p[s].y };
    int64_t uu = u.x * u.x + u.y * u.y;
    /* second axis: nearest point whose direction makes 30..150 degrees with u */
    int n2 = -1;
    int64_t bd2 = 0;
    for (int k = 0; k < n; ++k) {
      if (k == s || k == n1) continue;
      v2 wv = { p[k].x - p[s].x, p[k].y - p[s].y };
      int64_t cr = u.x * wv.y - u.y * wv.x;
      int64_t ww = wv.x * wv.x + wv.y * wv.y;
      if (4 * cr * cr < uu * ww) continue;
      if (n2 < 0 || ww < bd2) { n2 = k; bd2 = ww; }
    }
    if (n2 < 0) continue;
    v2 v = { p[n2].x - p[s].x, p[n2].y - p[s].y };

    int16_t lab[GW][GW];
    for (int i = 0; i < GW; ++i) for (int j = 0; j < GW; ++j) lab[i][j] = -1;
#define LAB(i, j) lab[(i) + GM][(j) + GM]
    int qi[GMAXPTS], qj[GMAXPTS];
    int qh = 0, qt = 0;
    LAB(0, 0) = (int16_t)s;  qi[qt] = 0; qj[qt] = 0; ++qt;
    LAB(1, 0) = (int16_t)n1; qi[qt] = 1; qj[qt] = 0; ++qt; used[n1] = 1;
    LAB(0, 1) = (int16_t)n2; qi[qt] = 0; qj[qt] = 1; ++qt; used[n2] = 1;
    int L = 3;
    static const int DI[4] = { 1, -1, 0, 0 }, DJ[4] = { 0, 0, 1, -1 };
    while (qh < qt) {
      int i = qi[qh], j = qj[qh];
      ++qh;
      v2 a = p[LAB(i, j)];
      for (int d = 0; d < 4; ++d) {
        int di = DI[d], dj = DJ[d];
        int ti = i + di, tj = j + dj;
        if (ti < -GM + 1 || ti > GM - 1 || tj < -GM + 1 || tj > GM - 1) continue;
        if (LAB(ti, tj) >= 0) continue;
        v2 pred;
        int64_t step2;
        int have = 0;
        if (LAB(i - di, j - dj) >= 0) {                 /* 1: extrapolate along the line */
          v2 b = p[LAB(i - di, j - dj)];
          pred.x = 2 * a.x - b.x; pred.y = 2 * a.y - b.y;
          step2 = d2(a, b);
          have = 1;
        }
        if (!have) {                                     /* 2: parallel edge one lattice line over */
          for (int o = -1; o <= 1 && !have; o += 2) {
            int oi = di ? 0 : o, oj = di ? o : 0;
            if (LAB(i + oi, j + oj) >= 0 && LAB(i + oi + di, j + oj + dj) >= 0) {
              v2 c0 = p[LAB(i + oi, j + oj)], c1 = p[LAB(i + oi + di, j + oj + dj)];
              pred.x = a.x + (c1.x - c0.x); pred.y = a.y + (c1.y - c0.y);
              step2 = d2(c0, c1);
              have = 1;
            }
          }
        }
        if (!have) {                                     /* 3: the seed's basis */
          v2 e = di ? u : v;
          int sg = di ? di : dj;
          pred.x = a.x + sg * e.x; pred.y = a.y + sg * e.y;
          step2 = e.x * e.x + e.y * e.y;
        }
        int64_t dist;
        int k = nearest_free(p, n, used, pred, &dist);
        if (k < 0) continue;
        if (8 * dist > step2) continue;                  /* within ~0.35 of the local step */
        LAB(ti, tj) = (int16_t)k;
        used[k] = 1;
        qi[qt] = ti; qj[qt] = tj; ++qt;
        ++L;
      }
    }
    if (L < need || L > need + GEXTRA) continue;
    int li[GMAXPTS], lj[GMAXPTS], lk[GMAXPTS], nl = 0;
    for (int i = -GM; i <= GM; ++i)
      for (int j = -GM; j <= GM; ++j)
        if (LAB(i, j) >= 0) { li[nl] = i; lj[nl] = j; lk[nl] = LAB(i, j); ++nl; }
    static const int SHEAR[7] = { 0, 1, -1, 2, -2, 3, -3 };
    int found = 0, transpose = 0, imin = 0, jmin = 0, shear = 0;
    if (L == need) {
      /* The second seed axis may be a lattice diagonal (v + k*u) in a foreshortened view: the
       * labelled set is then a sheared rectangle.  Undo the shear: first k in 0,1,-1,2,-2,3,-3 for
       * which (i + k*j, j) fills a cols x rows (or rows x cols) box. */
      for (int si2 = 0; si2 < 7 && !found; ++si2) {
        int k = SHEAR[si2];
        int i0 = 1 << 20, i1 = -(1 << 20), j0 = 1 << 20, j1 = -(1 << 20);
        for (int q = 0; q < nl; ++q) {
          int ii = li[q] + k * lj[q], jj = lj[q];
          if (ii < i0) i0 = ii;
          if (ii > i1) i1 = ii;
          if (jj < j0) j0 = jj;
          if (jj > j1) j1 = jj;
        }
        int bw = i1 - i0 + 1, bh = j1 - j0 + 1;
        if (bw == cols && bh == rows) { found = 1; transpose = 0; }
        else if (bw == rows && bh == cols) { found = 1; transpose = 1; }
        if (found) { imin = i0; jmin = j0; shear = k; }
      }
    } else {
      /* [B] round 4: MORE labels than the board has corners -- something junction-like next to the board (an object touching its
       * border squares) continued a row or a column by a cell or two (at most GEXTRA labels more: beyond, the seed is given up).  The board is then the one fully labelled cols x rows (or
       * rows x cols) window of the labelled set: shears in the same order; the first shear under which any window is full decides
       * -- exactly one window: taken, the labels outside it are dropped; several (a whole extra row or column): the seed is given
       * up.  (bbox == board and L == need above is the same statement for a set without extras.) */
      for (int si2 = 0; si2 < 7 && !found; ++si2) {
        int k = SHEAR[si2];
        int i0 = 1 << 20, i1 = -(1 << 20), j0 = 1 << 20, j1 = -(1 << 20);
        for (int q = 0; q < nl; ++q) {
          int ii = li[q] + k * lj[q], jj = lj[q];
          if (ii < i0) i0 = ii;
          if (ii > i1) i1 = ii;
          if (jj < j0) j0 = jj;
          if (jj > j1) j1 = jj;
        }
        int wins = 0, wi = 0, wj = 0, wt = 0;
        for (int tr = 0; tr < (cols == rows ? 1 : 2); ++tr) {
          int cw = tr ? rows : cols, ch = tr ? cols : rows;
          for (int a0 = i0; a0 + cw - 1 <= i1; ++a0)
            for (int b0 = j0; b0 + ch - 1 <= j1; ++b0) {
              int full = 1;
              for (int b = 0; b < ch && full; ++b)
                for (int a = 0; a < cw; ++a) {
                  int jj = b0 + b, oi = a0 + a - k * jj;
                  if (oi < -GM || oi > GM || LAB(oi, jj) < 0) { full = 0; break; }
                }
              if (full) { if (!wins) { wi = a0; wj = b0; wt = tr; } ++wins; }
            }
        }
        if (wins == 1) { found = 1; transpose = wt; imin = wi; jmin = wj; shear = k; }
        else if (wins > 1) break;
      }
    }
    if (!found) continue;
    /* bbox area == need and L == need and the shear is a bijection => every cell is filled */
    int32_t tmp[GMAXPTS];
    for (int q = 0; q < nl; ++q) {
      int a = li[q] + shear * lj[q] - imin, b = lj[q] - jmin;   /* a along the box width */
      if (a < 0 || b < 0 || a >= (transpose ? rows : cols) || b >= (transpose ? cols : rows)) continue;   /* a label outside the window (L > need) */
      int c = transpose ? b : a, r = transpose ? a : b;
      tmp[r * cols + c] = lk[q];
    }
    v2 P00 = p[tmp[0]], Pc = p[tmp[cols - 1]], Pr = p[tmp[(rows - 1) * cols]];
    int64_t cr = (Pc.x - P00.x) * (Pr.y - P00.y) - (Pc.y - P00.y) * (Pr.x - P00.x);
    int flipc = cr < 0;
    int32_t t2[GMAXPTS];
    for (int r = 0; r < rows; ++r)
      for (int c = 0; c < cols; ++c) t2[r * cols + c] = tmp[r * cols + (flipc ? cols - 1 - c : c)];
    v2 A0 = p[t2[0]], A1 = p[t2[need - 1]];
    int rot = (A1.y < A0.y) || (A1.y == A0.y && A1.x < A0.x);
    for (int k = 0; k < need; ++k) order_out[k] = rot ? t2[need - 1 - k] : t2[k];
    return 1;
#undef LAB
  }
  return 0;
}

/* object points of the board's inner corners, index = row*cols + col (see above) */
void orc_board_object_points(int cols, int rows, double square, double* obj)
{
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) {
      double* o = obj + 3 * (r * cols + c);
      o[0] = ((double)c - 0.5 * (double)(cols - 1)) * square;
      o[1] = (0.5 * (double)(rows - 1) - (double)r) * square;
      o[2] = 0.0;
    }
}
