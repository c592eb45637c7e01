/*
 * orc_fiducial.c -- CPU oracle, square-fiducial form of stages a4 (quad extraction) and a6 (decode).
 * TEST INFRASTRUCTURE.
 *
 * What the reference consumes from this stage, per tag: id[0], size[0], four pixel corners
 * (real_preprocessing/src/corner_detections.cpp:48-54) in the order bl,br,tr,tl with object points
 * (+-size/2, +-size/2, 0) (real_preprocessing/src/camera_pose.cpp:152-161).  The reference gets them
 * from the external apriltag_ros/apriltag packages (README.md:15-16,65), which are not in this image;
 * SURVEY.md appendix C restates that pipeline (union-find segmentation, boundary clustering, quad
 * fit, decode).  [B] This build extracts quads differently, on top of the stages it already has
 * (threshold map, Harris candidates, sub-pixel refinement): convex black corners are classified on
 * the threshold map, linked along black/white boundaries, 4-cycles are quads, and the decode follows
 * appendix C.5 in spirit (homography, cell sampling, black/white levels, family lookup with <= 2 bit
 * errors over 4 rotations).  All decisions are integer (fixed-point sampling) given the refined
 * corner positions, whose arithmetic is bit-reproducible (a5).
 *
 * Tag layout [B]: 8x8 cells, one-cell black border, 6x6 payload (white = 1), code word = payload
 * row-major, MSB first.  The family table is data (rcc_config.family_codes).
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "orc.h"

static const int8_t RING16F[16][2] = {
  { 5, 0}, { 5, 2}, { 4, 4}, { 2, 5}, { 0, 5}, {-2, 5}, {-4, 4}, {-5, 2},
  {-5, 0}, {-5,-2}, {-4,-4}, {-2,-5}, { 0,-5}, { 2,-5}, { 4,-4}, { 5,-2}
};

/* convex black corner test at integer pixel (x,y) on the GREY image with a local threshold (the
 * tile threshold map is 127 in flat areas, and the quiet zone around a tag is flat): 16-sample
 * radius-5 ring, t = (min+max)/2 of the ring, contrast >= min_contrast; exactly one black arc
 * (samples <= t) of 2..7 samples.  d1/d2: boundary directions at the arc start / end (sums of the
 * two ring offsets that straddle the transition).  *thr receives t. */
int orc_fid_corner_class(const uint8_t* g, int w, int h, int x, int y, int min_contrast, int d1[2], int d2[2], int* thr)
{
  if (x < 5 || y < 5 || x >= w - 5 || y >= h - 5) return 0;
  int v[16], lo = 255, hi = 0;
  for (int k = 0; k < 16; ++k) {
    v[k] = g[(size_t)(y + RING16F[k][1]) * w + (x + RING16F[k][0])];
    if (v[k] < lo) lo = v[k];
    if (v[k] > hi) hi = v[k];
  }
  if (hi - lo < min_contrast) return 0;
  const int t = (lo + hi) / 2;
  for (int k = 0; k < 16; ++k) v[k] = v[k] > t;
  int tr = 0, a = -1;
  for (int k = 0; k < 16; ++k) {
    if (v[k] != v[(k + 1) & 15]) ++tr;
    if (v[k] == 1 && v[(k + 1) & 15] == 0) a = (k + 1) & 15;
  }
  if (tr != 2 || a < 0) return 0;
  int len = 0;
  while (len < 16 && v[(a + len) & 15] == 0) ++len;
  if (len < 2 || len > 7) return 0;
  const int am = (a + 15) & 15, e = (a + len - 1) & 15, en = (a + len) & 15;
  d1[0] = RING16F[am][0] + RING16F[a][0]; d1[1] = RING16F[am][1] + RING16F[a][1];
  d2[0] = RING16F[e][0] + RING16F[en][0]; d2[1] = RING16F[e][1] + RING16F[en][1];
  *thr = t;
  return 1;
}

static int rdiv10(int v) { return (v * 3 + (v >= 0 ? 5 : -5)) / 10; }

/* is the straight segment from corner i to corner j a boundary with black on the n side? */
static int edge_ok(const uint8_t* g, int w, int h, int xi, int yi, int wx, int wy, int nx, int ny, int t)
{
  const int ox = rdiv10(nx), oy = rdiv10(ny);
  int good = 0;
  for (int k = 0; k < 8; ++k) {
    int mx = (xi * 16 + wx * (2 * k + 1) + 8) >> 4, my = (yi * 16 + wy * (2 * k + 1) + 8) >> 4;
    int bx = mx + ox, by = my + oy, cx = mx - ox, cy = my - oy;
    if (bx < 0 || by < 0 || bx >= w || by >= h || cx < 0 || cy < 0 || cx >= w || cy >= h) return 0;
    if (g[(size_t)by * w + bx] <= t && g[(size_t)cy * w + cx] > t) ++good;
  }
  return good >= 7;
}

/* homography P_k -> q_k for the tag's cell frame: (0,0),(8,0),(8,8),(0,8); 8x8 system, partial
 * pivoting, fixed operation order (one rounded IEEE op per step) */
int orc_fid_homography(const double q[8], double H[9])
{
  static const double P[8] = { 0, 0, 8, 0, 8, 8, 0, 8 };
  double M[8][9];
  for (int k = 0; k < 4; ++k) {
    const double u = P[2 * k], v = P[2 * k + 1], x = q[2 * k], y = q[2 * k + 1];
    double* a = M[2 * k];
    double* b = M[2 * k + 1];
    a[0] = u; a[1] = v; a[2] = 1; a[3] = 0; a[4] = 0; a[5] = 0; a[6] = -(u * x); a[7] = -(v * x); a[8] = x;
    b[0] = 0; b[1] = 0; b[2] = 0; b[3] = u; b[4] = v; b[5] = 1; b[6] = -(u * y); b[7] = -(v * y); b[8] = y;
  }
  for (int c = 0; c < 8; ++c) {
    int piv = c;
    double best = fabs(M[c][c]);
    for (int r = c + 1; r < 8; ++r) if (fabs(M[r][c]) > best) { best = fabs(M[r][c]); piv = r; }
    if (!(best > 1e-12)) return 0;
    if (piv != c) for (int k = 0; k < 9; ++k) { double t = M[c][k]; M[c][k] = M[piv][k]; M[piv][k] = t; }
    for (int r = c + 1; r < 8; ++r) {
      const double f = M[r][c] / M[c][c];
      for (int k = c; k < 9; ++k) { double t = f * M[c][k]; M[r][k] = M[r][k] - t; }
    }
  }
  for (int r = 7; r >= 0; --r) {
    double s = M[r][8];
    for (int k = r + 1; k < 8; ++k) { double t = M[r][k] * H[k]; s = s - t; }
    H[r] = s / M[r][r];
  }
  H[8] = 1.0;
  return 1;
}

/* grey at tag-frame point (u,v) through H: fixed-point (1/16 px) bilinear; -1 if outside the image */
static int sample_cell(const uint8_t* g, int w, int h, const double H[9], double u, double v)
{
  double a = H[0] * u, b = H[1] * v; double px = a + b; px = px + H[2];
  a = H[3] * u; b = H[4] * v; double py = a + b; py = py + H[5];
  a = H[6] * u; b = H[7] * v; double pw = a + b; pw = pw + H[8];
  px = px / pw; py = py / pw;
  if (!(px >= 0.0 && py >= 0.0 && px <= (double)(w - 2) && py <= (double)(h - 2))) return -1;
  int X = (int)rint(px * 16.0), Y = (int)rint(py * 16.0);
  int ix = X >> 4, iy = Y >> 4, fx = X & 15, fy = Y & 15;
  if (ix < 0 || iy < 0 || ix >= w - 1 || iy >= h - 1) return -1;
  const uint8_t* p = g + (size_t)iy * w + ix;
  int acc = (16 - fx) * (16 - fy) * p[0] + fx * (16 - fy) * p[1] + (16 - fx) * fy * p[w] + fx * fy * p[w + 1];
  return (acc + 128) >> 8;
}

static uint64_t rot36(uint64_t c)
{
  uint64_t o = 0;
  for (int r = 0; r < 6; ++r)
    for (int cc = 0; cc < 6; ++cc) {
      uint64_t b = (c >> (35 - (cc * 6 + (5 - r)))) & 1u;       /* M'[r][c] = M[c][5-r] */
      o |= b << (35 - (r * 6 + cc));
    }
  return o;
}

/* decode the quad q (4 corners, clockwise on screen) -> id / hamming / rotation; returns 1 on success */
int orc_fid_decode(const uint8_t* g, int w, int h, const double q[8], const uint64_t* codes, int ncodes,
                   int max_hamming, int* id_out, int* ham_out, int* rot_out)
{
  double H[9];
  if (!orc_fid_homography(q, H)) return 0;
  int cell[8][8];
  int bsum = 0, wsum = 0;
  for (int r = 0; r < 8; ++r)
    for (int c = 0; c < 8; ++c) {
      int s = sample_cell(g, w, h, H, (double)c + 0.5, (double)r + 0.5);
      if (s < 0) return 0;
      cell[r][c] = s;
      if (r == 0 || r == 7 || c == 0 || c == 7) bsum += s;
    }
  for (int i = -1; i <= 8; ++i) {           /* quiet zone ring, one cell out: 36 samples */
    int s0 = sample_cell(g, w, h, H, (double)i + 0.5, -0.5), s1 = sample_cell(g, w, h, H, (double)i + 0.5, 8.5);
    if (s0 < 0 || s1 < 0) return 0;
    wsum += s0 + s1;
    if (i >= 0 && i <= 7) {
      int s2 = sample_cell(g, w, h, H, -0.5, (double)i + 0.5), s3 = sample_cell(g, w, h, H, 8.5, (double)i + 0.5);
      if (s2 < 0 || s3 < 0) return 0;
      wsum += s2 + s3;
    }
  }
  const int black = bsum / 28, white = wsum / 36;
  if (white - black < 40) return 0;
  const int thr = (black + white) / 2;
  for (int r = 0; r < 8; ++r)
    for (int c = 0; c < 8; ++c)
      if ((r == 0 || r == 7 || c == 0 || c == 7) && cell[r][c] >= thr) return 0;
  uint64_t S = 0;
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c)
      if (cell[r + 1][c + 1] > thr) S |= (uint64_t)1 << (35 - (r * 6 + c));
  int best_id = -1, best_h = 99, best_rot = 0;
  uint64_t M = S;
  for (int rot = 0; rot < 4; ++rot) {
    for (int k = 0; k < ncodes; ++k) {
      int hd = __builtin_popcountll(M ^ codes[k]);
      if (hd < best_h) { best_h = hd; best_id = k; best_rot = rot; }    /* first best wins: rotation, then id */
    }
    M = rot36(M);
  }
  if (best_id < 0 || best_h > max_hamming) return 0;
  *id_out = best_id; *ham_out = best_h; *rot_out = best_rot;
  return 1;
}

/* ---- corner refinement of a quad in the refine_edges form (SURVEY.md appendix C.4 [U]: per edge >= 16 samples; at each,
 * a scan along the normal in 0.25-px steps, weights (g2 - g1)^2 of the correct polarity, weighted-mean offset; a total-
 * least-squares line through the refined points; corners = intersections of adjacent lines).  Restated [B] so that every
 * decision is an integer and every floating-point step one rounded IEEE operation (+ - * / sqrt, -ffp-contract=off):
 *   qi        the quad's integer corners (x, y), clockwise on screen; black lies on the (-dy, dx) side of every edge a -> b
 *   samples   16 per edge at a + alpha (b - a), alpha = (s + 2) / 19, s = 0..15 (the two positions nearest either corner are
 *             left out: there the other edge bends the profile)
 *   scan      offsets k / 2 px along the OUTWARD normal n = (dy, -dx) / |d|, k = -6 .. 6 (half-pixel steps measure the same
 *             corner error against the renderer's ground truth as appendix C.4's quarter-pixel steps, at half the samples);
 *             g1 = the image one pixel further
 *             out, g2 = one pixel further in, both sampled bilinearly in 1/16-px fixed point as 256 x grey (bil16; a step
 *             whose samples leave the image is skipped); weight (g1 - g2)^2 where g1 > g2 (white outside), else 0;
 *             offset = (sum k w / sum w) / 2  -- integer sums
 *   line      moments of the refined points relative to a, summed over the 16 samples along the pairing tree
 *             v[l] += v[l ^ 8], ^ 4, ^ 2, ^ 1 (what a 16-lane butterfly does); centroid E, covariance C; the normal is the
 *             eigenvector of C's smaller eigenvalue, taken in its well-conditioned form; fewer than 4 valid samples or a
 *             vanishing normal: the line through the integer corners
 *   corners   corner c = intersection of the lines of edges c - 1 and c, solved relative to qi[c]; kept at the integer
 *             corner if the lines are (nearly) parallel or the intersection lies more than 4 px away from it */
/* 256 x grey at (px, py), bilinear with the position rounded to 1/16 px (0 <= px <= w - 2, 0 <= py <= h - 2) */
static int bil16(const uint8_t* g, int w, double px, double py)
{
  const int X = (int)rint(px * 16.0), Y = (int)rint(py * 16.0);
  const int ix = X >> 4, iy = Y >> 4, fx = X & 15, fy = Y & 15;
  const uint8_t* p = g + (size_t)iy * w + ix;
  return (16 - fx) * (16 - fy) * p[0] + fx * (16 - fy) * p[1] + (16 - fx) * fy * p[w] + fx * fy * p[w + 1];   /* 256 x grey */
}
static double tree16(double* v)
{
  for (int off = 8; off >= 1; off >>= 1) {
    double nv[16];
    for (int l = 0; l < 16; ++l) nv[l] = v[l] + v[l ^ off];
    for (int l = 0; l < 16; ++l) v[l] = nv[l];
  }
  return v[0];
}

static void refine_edges_pass(const uint8_t* g, int w, int h, const double qi[8], double qr[8])
{
  double E[4][2], V[4][2];         /* per edge: a point of the line (absolute) and its normal (any length) */
  for (int e = 0; e < 4; ++e) {
    const double ax = qi[2 * e], ay = qi[2 * e + 1], bx = qi[2 * ((e + 1) & 3)], by = qi[2 * ((e + 1) & 3) + 1];
    const double dx = bx - ax, dy = by - ay;
    const double dxx = dx * dx, dyy = dy * dy;
    const double L = sqrt(dxx + dyy);
    double nx = 0.0, ny = 0.0;
    if (L > 0.0) { nx = dy / L; ny = -dx / L; }
    double sx[16], sy[16], sxx[16], sxy[16], syy[16], sn[16];
    for (int s = 0; s < 16; ++s) {
      const double alpha = (double)(s + 2) / 19.0;
      const double tx = alpha * dx, ty = alpha * dy;
      const double x0 = ax + tx, y0 = ay + ty;
      long long Mn = 0, Mc = 0;
      for (int k = -6; k <= 6; ++k) {
        const double t1 = (double)(k + 2) * 0.5, t2 = (double)(k - 2) * 0.5;
        const double u1 = t1 * nx, v1 = t1 * ny, u2 = t2 * nx, v2 = t2 * ny;
        const double x1 = x0 + u1, y1 = y0 + v1, x2 = x0 + u2, y2 = y0 + v2;
if (!(x1 >= 0.0 && y1 >= 0.0 && x1 <= (double)(w - 2) && y1 <= (double)(h - 2) &&
              x2 >= 0.0 && y2 >= 0.0 && x2 <= (double)(w - 2) && y2 <= (double)(h - 2))) continue;
        const int g1 = bil16(g, w, x1, y1), g2 = bil16(g, w, x2, y2);
        if (g1 <= g2) continue;
        const long long wt = (long long)(g1 - g2) * (g1 - g2);
        Mn += wt * k; Mc += wt;
      }
      if (Mc == 0 || !(L > 0.0)) { sx[s] = sy[s] = sxx[s] = sxy[s] = syy[s] = sn[s] = 0.0; continue; }
      const double n0 = ((double)Mn / (double)Mc) * 0.5;
      const double ox = n0 * nx, oy = n0 * ny;
      const double rx = tx + ox, ry = ty + oy;          /* refined point relative to a */
      sx[s] = rx; sy[s] = ry; sxx[s] = rx * rx; sxy[s] = rx * ry; syy[s] = ry * ry; sn[s] = 1.0;
    }
    const double N = tree16(sn), Sx = tree16(sx), Sy = tree16(sy), Sxx = tree16(sxx), Sxy = tree16(sxy), Syy = tree16(syy);
    E[e][0] = ax; E[e][1] = ay; V[e][0] = nx; V[e][1] = ny;      /* fallback: the edge as given */
    if (N >= 4.0) {
      const double Ex = Sx / N, Ey = Sy / N;
      const double Cxx = Sxx / N - Ex * Ex, Cxy = Sxy / N - Ex * Ey, Cyy = Syy / N - Ey * Ey;
      const double hd = (Cxx - Cyy) * 0.5;
      const double r = sqrt(hd * hd + Cxy * Cxy);
      double vx, vy;
      if (hd >= 0.0) { vx = Cxy; vy = -hd - r; } else { vx = hd - r; vy = Cxy; }
      const double vv = vx * vx + vy * vy;
      if (vv > 1e-12) { E[e][0] = ax + Ex; E[e][1] = ay + Ey; V[e][0] = vx; V[e][1] = vy; }
    }
  }
  for (int c = 0; c < 4; ++c) {
    const int a = (c + 3) & 3, b = c;           /* edges a (arriving at corner c) and b (leaving it) */
    const double cx = qi[2 * c], cy = qi[2 * c + 1];
    const double ea = V[a][0] * (E[a][0] - cx) + V[a][1] * (E[a][1] - cy);
    const double eb = V[b][0] * (E[b][0] - cx) + V[b][1] * (E[b][1] - cy);
    const double det = V[a][0] * V[b][1] - V[a][1] * V[b][0];
    const double na = V[a][0] * V[a][0] + V[a][1] * V[a][1], nb = V[b][0] * V[b][0] + V[b][1] * V[b][1];
    double px = 0.0, py = 0.0;
    int ok = det * det > 1e-6 * (na * nb);      /* sin^2 of the angle between the lines > 1e-6 */
    if (ok) {
      px = (ea * V[b][1] - eb * V[a][1]) / det;
      py = (V[a][0] * eb - V[b][0] * ea) / det;
      ok = (px * px + py * py <= 16.0);
    }
    qr[2 * c] = ok ? cx + px : cx;
    qr[2 * c + 1] = ok ? cy + py : cy;
  }
}

void orc_fid_refine_edges(const uint8_t* g, int w, int h, const int qi[8], double qr[8])
{
  double q0[8];
  for (int k = 0; k < 8; ++k) q0[k] = (double)qi[k];
  refine_edges_pass(g, w, h, q0, qr);
}

int orc_fid_detect_dbg(const uint8_t* grey, int w, int h, int min_contrast, const orc_cand* pre, const double* xy,
                       int n, const uint64_t* codes, int ncodes, int max_hamming, int refine_mode, rcc_detection* out, int cap,
                       int32_t* dbg_ok, int32_t* dbg_nxt);

/* whole fiducial stage for one frame.  pre: suppressed candidate list (sorted by (y,x)), xy: their
 * refined positions.  Writes up to cap records {id, hamming, corners bl,br,tr,tl}; returns the count. */
int orc_fid_detect(const uint8_t* grey, int w, int h, int min_contrast, const orc_cand* pre, const double* xy,
                   int n, const uint64_t* codes, int ncodes, int max_hamming, int refine_mode, rcc_detection* out, int cap)
{
  return orc_fid_detect_dbg(grey, w, h, min_contrast, pre, xy, n, codes, ncodes, max_hamming, refine_mode, out, cap, NULL, NULL);
}

/* Corners are classified and linked at the rounded positions xy (the a5 pass: coarse for RCC_TAG_REFINE_EDGES, full for
 * RCC_TAG_REFINE_CORNER_SUBPIX).  EDGES: each quad's corners come from orc_fid_refine_edges started at those rounded
 * positions; CORNER_SUBPIX: the refined positions themselves are the reported corners. */
int orc_fid_detect_dbg(const uint8_t* grey, int w, int h, int min_contrast, const orc_cand* pre, const double* xy,
                       int n, const uint64_t* codes, int ncodes, int max_hamming, int refine_mode, rcc_detection* out, int cap,
                       int32_t* dbg_ok, int32_t* dbg_nxt)
{
  const int edges = (refine_mode == RCC_TAG_REFINE_EDGES);
  if (n > RCC_MAX_KEPT_FIDUCIAL) n = RCC_MAX_KEPT_FIDUCIAL;
  int* px = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1) * 7);
  int* py = px + n; int* ok = py + n; int* d1x = ok + n; int* d1y = d1x + n; int* nxt = d1y + n;
  int* thr = nxt + n;
  for (int i = 0; i < n; ++i) {
    px[i] = (int)floor(xy[2 * i] + 0.5); py[i] = (int)floor(xy[2 * i + 1] + 0.5);
    int a[2], b[2];
    thr[i] = 0;
    ok[i] = orc_fid_corner_class(grey, w, h, px[i], py[i], min_contrast, a, b, &thr[i]);
    d1x[i] = a[0]; d1y[i] = a[1];
    nxt[i] = -1;
  }
  /* link along d1: black lies on the n1 = (-d1y, d1x) side of the direction of travel */
  for (int i = 0; i < n; ++i) {
    if (!ok[i]) continue;
    const long long dx = d1x[i], dy = d1y[i], dd = dx * dx + dy * dy;
    long long bestd = 0;
    int best = -1;
    for (int j = 0; j < n; ++j) {
      if (j == i || !ok[j]) continue;
      const long long wx = px[j] - px[i], wy = py[j] - py[i], ww = wx * wx + wy * wy;
      if (ww < 64) continue;
      if (wx * dx + wy * dy <= 0) continue;
      const long long cr = wx * dy - wy * dx;
      if (8 * cr * cr > ww * dd) continue;          /* within ~20 degrees of the ring-quantised direction; edge_ok decides */
      if (best >= 0 && ww >= bestd) continue;
      if (!edge_ok(grey, w, h, px[i], py[i], (int)wx, (int)wy, (int)-dy, (int)dx, thr[i])) continue;
      best = j; bestd = ww;
    }
    nxt[i] = best;
  }
  if (dbg_ok) for (int i = 0; i < n; ++i) { dbg_ok[i] = ok[i]; dbg_nxt[i] = nxt[i]; }
  int m = 0;
  for (int i = 0; i < n; ++i) {
    if (!ok[i]) continue;
    int j = nxt[i]; if (j < 0) continue;
    int k = nxt[j]; if (k < 0) continue;
    int l = nxt[k]; if (l < 0) continue;
    if (nxt[l] != i) continue;
    if (j == k || j == l || k == l || k == i || j == i || l == i) continue;
    if (!(i < j && i < k && i < l)) continue;                  /* emit each cycle once, from its smallest index */
    const int idx[4] = { i, j, k, l };
    double q[8];
    if (edges) {
      int qi[8];
      for (int c = 0; c < 4; ++c) { qi[2 * c] = px[idx[c]]; qi[2 * c + 1] = py[idx[c]]; }
      const long long cri = (long long)(qi[2] - qi[0]) * (qi[5] - qi[3]) - (long long)(qi[3] - qi[1]) * (qi[4] - qi[2]);
      if (!(cri > 0)) continue;                                   /* clockwise on screen (on the integer corners) */
      orc_fid_refine_edges(grey, w, h, qi, q);
    } else {
      for (int c = 0; c < 4; ++c) { q[2 * c] = xy[2 * idx[c]]; q[2 * c + 1] = xy[2 * idx[c] + 1]; }
    }
    const double cr = (q[2] - q[0]) * (q[5] - q[3]) - (q[3] - q[1]) * (q[4] - q[2]);
    if (!(cr > 0.0)) continue;                                    /* clockwise on screen */
    int id, ham, rot;
    if (!orc_fid_decode(grey, w, h, q, codes, ncodes, max_hamming, &id, &ham, &rot)) continue;
    if (m < cap) {
      rcc_detection* d = out + m;
      memset(d, 0, sizeof(*d));
      d->id = id; d->hamming = ham; d->ncorners = 4;
      /* true top-left = q[rot]; reference order bl, br, tr, tl */
      const int tl = rot & 3, trc = (rot + 1) & 3, br = (rot + 2) & 3, bl = (rot + 3) & 3;
      const int ord[4] = { bl, br, trc, tl };
      for (int c = 0; c < 4; ++c) { d->corners[c][0] = q[2 * ord[c]]; d->corners[c][1] = q[2 * ord[c] + 1]; }
    }
    ++m;
  }
  free(px);
  return m;
}
