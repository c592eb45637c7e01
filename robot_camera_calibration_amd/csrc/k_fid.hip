// k_fid.hip -- square-fiducial form of stages a4 (quad extraction) and a6 (decode), then a7 per tag.
//
// What the reference consumes per tag: id[0], size[0], four pixel corners
// (real_preprocessing/src/corner_detections.cpp:48-54), order bl,br,tr,tl, object points
// (+-size/2, +-size/2, 0) (real_preprocessing/src/camera_pose.cpp:152-161); it receives them from the
// external apriltag packages (README.md:15-16,65).  Definitions: DESIGN.md section 3 (a4/a6 fiducial
// form) -- convex black corners classified on the grey image with a local threshold, linked along
// black/white boundaries, 4-cycles are quads, decode by homography + fixed-point cell sampling +
// family lookup (<= max_hamming errors, 4 rotations).  Every decision is integer given the refined
// corner positions, and the fp64 homography solve is written one rounded operation per step
// (-ffp-contract=off), so the outputs equal the oracle's bit for bit.
//
// One 256-thread block per frame (39.9 KB of LDS: four blocks per CU): O(n^2) integer pair gating over the classified
// corners, 16-pixel edge probes for the pairs that pass, a handful of decodes -- VALU-bound once ~1000 frames are
// resident, never bandwidth-bound; frames are the parallel axis.  DESIGN.md section 5 has the history.
#include "rcc_internal.h"
// The solver routines of pnp_core.h are inlined (no RCC_PNP_NOINLINE): round 1 kept them out of line after a suspected
// hipcc -O3 miscompile that round 2 could not reproduce -- the fully inlined -O3 build passes every pose parity test
// (the one recorded failure was the test's own: the Rodrigues round trip is not unique beyond |r| = pi) and is faster
// (24 456 tag poses 0.45 -> 0.31 ms, board pose 0.27 -> 0.25 ms; profiles/r02_e_pnp_inline.txt).
#include "pnp_core.h"

#define FID_MAXN RCC_MAX_KEPT_FIDUCIAL

__constant__ int8_t c_ring16f[16][2] = {
  { 5, 0}, { 5, 2}, { 4, 4}, { 2, 5}, { 0, 5}, {-2, 5}, {-4, 4}, {-5, 2},
  {-5, 0}, {-5,-2}, {-4,-4}, {-2,-5}, { 0,-5}, { 2,-5}, { 4,-4}, { 5,-2}
};

__device__ __forceinline__ int fid_rdiv10(int v) { return (v * 3 + (v >= 0 ? 5 : -5)) / 10; }

__device__ bool fid_corner_class(const uint8_t* __restrict__ g, int w, int h, int x, int y, int min_contrast,
                                 int& d1x, int& d1y, int& thr)
{
  if (x < 5 || y < 5 || x >= w - 5 || y >= h - 5) return false;
  int v[16], lo = 255, hi = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    v[k] = g[(size_t)(y + c_ring16f[k][1]) * w + (x + c_ring16f[k][0])];
    lo = min(lo, v[k]); hi = max(hi, v[k]);
  }
  if (hi - lo < min_contrast) return false;
  const int t = (lo + hi) / 2;
  unsigned bits = 0;                         // bit k = ring sample k is white
#pragma unroll
  for (int k = 0; k < 16; ++k) bits |= (unsigned)(v[k] > t) << k;
  const unsigned rotl = ((bits << 1) | (bits >> 15)) & 0xFFFFu;   // bit k = sample k-1
  const unsigned diff = bits ^ rotl;                              // bit k: sample k-1 != sample k
  if (__popc(diff) != 2) return false;
  // a = first black sample after a white one: sample a-1 white (rotl bit a set), sample a black
  const unsigned start = rotl & ~bits & 0xFFFFu;
  if (start == 0) return false;
  const int a = __ffs(start) - 1;
  int len = 0;
  while (len < 16 && !((bits >> ((a + len) & 15)) & 1u)) ++len;
  if (len < 2 || len > 7) return false;
  const int am = (a + 15) & 15;
  d1x = c_ring16f[am][0] + c_ring16f[a][0];
  d1y = c_ring16f[am][1] + c_ring16f[a][1];
  thr = t;
  return true;
}

// The segment test of the link stage: 8 two-sided probes along the segment (black on the n side, white on the other), good
// if no probe leaves the image and at least 7 of the 8 are right.  Evaluated in two parts -- first the probes of FID_S1,
// then the rest -- because almost every pair the link stage tries is wrong and fails both probes of the first part, after
// which 7 of 8 cannot be reached: a quarter of the (scattered, one byte per lane) loads decides most pairs.  Within a
// part all loads are issued before any is tested.
template <unsigned MASK>
__device__ __forceinline__ void fid_edge_part(const uint8_t* __restrict__ g, int w, int h, int xi, int yi, int wx, int wy, int ox, int oy, int t,
                                              bool& inb, int& good)
{
  int vb[8], vc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (!((MASK >> k) & 1u)) continue;
    const int mx = (xi * 16 + wx * (2 * k + 1) + 8) >> 4, my = (yi * 16 + wy * (2 * k + 1) + 8) >> 4;
    const int bx = mx + ox, by = my + oy, cx = mx - ox, cy = my - oy;
    const bool ok = !(bx < 0 || by < 0 || bx >= w || by >= h || cx < 0 || cy < 0 || cx >= w || cy >= h);
    inb = inb && ok;
    const int bxc = min(max(bx, 0), w - 1), byc = min(max(by, 0), h - 1), cxc = min(max(cx, 0), w - 1), cyc = min(max(cy, 0), h - 1);
    vb[k] = g[(size_t)byc * w + bxc];
    vc[k] = g[(size_t)cyc * w + cxc];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if ((MASK >> k) & 1u) good += (vb[k] <= t && vc[k] > t) ? 1 : 0;
}
#define FID_S1 0x42u          // probes 1 and 6 of 8 (3/16 and 13/16 along the segment)
#define FID_S2 (0xFFu & ~FID_S1)
// first part: can the pair still pass?  (false: a probe left the image, or both probes are wrong -- at most 6 of 8 remain)
__device__ __forceinline__ bool fid_edge_first(const uint8_t* __restrict__ g, int w, int h, int xi, int yi, int wx, int wy, int nx, int ny, int t, int& good1)
{
  bool inb = true;
  good1 = 0;
  fid_edge_part<FID_S1>(g, w, h, xi, yi, wx, wy, fid_rdiv10(nx), fid_rdiv10(ny), t, inb, good1);
  return inb && good1 >= 1;
}
__device__ __forceinline__ bool fid_edge_rest(const uint8_t* __restrict__ g, int w, int h, int xi, int yi, int wx, int wy, int nx, int ny, int t, int good1)
{
  bool inb = true;
  int good = good1;
  fid_edge_part<FID_S2>(g, w, h, xi, yi, wx, wy, fid_rdiv10(nx), fid_rdiv10(ny), t, inb, good);
  return inb && good >= 7;
}

__device__ bool fid_homography(const double q[8], double H[9])
{
  const double P[8] = { 0, 0, 8, 0, 8, 8, 0, 8 };
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
    if (!(best > 1e-12)) return false;
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
  return true;
}

__device__ int fid_sample(const uint8_t* __restrict__ g, int w, int h, const double H[9], double u, double v)
{
  double a = H[0] * u, b = H[1] * v; double px = a + b; px = px + H[2];
  a = H[3] * u; b = H[4] * v; double py = a + b; py = py + H[5];
  a = H[6] * u; b = H[7] * v; double pw = a + b; pw = pw + H[8];
  px = px / pw; py = py / pw;
  if (!(px >= 0.0 && py >= 0.0 && px <= (double)(w - 2) && py <= (double)(h - 2))) return -1;
  const int X = (int)rint(px * 16.0), Y = (int)rint(py * 16.0);
  const int ix = X >> 4, iy = Y >> 4, fx = X & 15, fy = Y & 15;
  if (ix < 0 || iy < 0 || ix >= w - 1 || iy >= h - 1) return -1;
  const uint8_t* p = g + (size_t)iy * w + ix;
  const int acc = (16 - fx) * (16 - fy) * p[0] + fx * (16 - fy) * p[1] + (16 - fx) * fy * p[w] + fx * fy * p[w + 1];
  return (acc + 128) >> 8;
}

__device__ uint64_t fid_rot36(uint64_t c)
{
  uint64_t o = 0;
  for (int r = 0; r < 6; ++r)
    for (int cc = 0; cc < 6; ++cc) {
      const uint64_t b = (c >> (35 - (cc * 6 + (5 - r)))) & 1u;
      o |= b << (35 - (r * 6 + cc));
    }
  return o;
}

// Decode of one quad by the G lanes of a group (G | 64; lanes lg = 0..G-1 all call this with the same q; the control flow
// is wave-uniform, so a group without a quad passes valid = false and comes back with false).  Definition (DESIGN.md
// section 3, a6): homography of the 8x8-cell tag square, 64 cell samples + 36 samples of the white ring around it (all
// must lie inside the image), black = border mean, white = ring mean, contrast >= 40, threshold at the midpoint, no white
// border cell, payload = the 36 inner cells, best of 4 rotations x ncodes code words by (hamming, rotation, id), accepted
// up to max_hamming.  One lane per quad made this 0.2-0.4 ms of the kernel (~100 dependent sample loads, a 64-entry
// scratch array, 4 * ncodes popcounts on a single lane); here every lane takes 100 / G samples and ncodes / G code words
// and the three decisions and the best key are reduced over the group: same operations per sample and per code word,
// hence the same result.
struct fid_hit { int16_t id; int8_t ham, rot; };      // id < 0: none

#define FID_QCAP 512                                   // pairs per wave: 64 lanes x 8 scan steps; 4 waves x 512 x 4 B = the s_hit array
static_assert(sizeof(fid_hit) == 4 && FID_MAXN * sizeof(fid_hit) >= 4 * FID_QCAP * sizeof(uint32_t), "pair queue aliases s_hit");
static_assert(FID_MAXN <= 2048, "11-bit compact corner index in the pair queue and the link key");

template <int G>
__device__ __forceinline__ int fid_gsum(int v) { for (int o = G >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, G); return v; }
template <int G>
__device__ __forceinline__ unsigned fid_gor(unsigned v) { for (int o = G >> 1; o > 0; o >>= 1) v |= (unsigned)__shfl_xor((int)v, o, G); return v; }
template <int G>
__device__ __forceinline__ unsigned fid_gmin(unsigned v) { for (int o = G >> 1; o > 0; o >>= 1) v = min(v, (unsigned)__shfl_xor((int)v, o, G)); return v; }

template <int G>
__device__ bool fid_decode_group(const uint8_t* __restrict__ g, int w, int h, const double q[8], const uint64_t* __restrict__ codes,
                                 int ncodes, int max_hamming, int lg, bool valid, int& id, int& ham, int& rot)
{
  double H[9];
  for (int k = 0; k < 9; ++k) H[k] = 0.0;
  const bool hok = valid && fid_homography(q, H);
  constexpr int NS = (100 + G - 1) / G;
  int sv[NS];
  int bsum = 0, wsum = 0;
  unsigned fail = hok ? 0u : 1u;
#pragma unroll
  for (int m = 0; m < NS; ++m) {
    const int s = lg + G * m;
    const bool cellp = s < 64, use = s < 100;
    double u, v;
    if (cellp) { u = (double)(s & 7) + 0.5; v = (double)(s >> 3) + 0.5; }
    else {
      const int t = s - 64;                                 // the white ring: top 10, bottom 10, left 8, right 8
      if (t < 10) { u = (double)(t - 1) + 0.5; v = -0.5; }
      else if (t < 20) { u = (double)(t - 11) + 0.5; v = 8.5; }
      else if (t < 28) { u = -0.5; v = (double)(t - 20) + 0.5; }
      else { u = 8.5; v = (double)(t - 28) + 0.5; }
    }
    const int val = (use && hok) ? fid_sample(g, w, h, H, u, v) : 0;
    if (val < 0) fail = 1u;
    sv[m] = val;
    const int r = s >> 3, c = s & 7;
    if (cellp && (r == 0 || r == 7 || c == 0 || c == 7)) bsum += val;
    if (!cellp && use) wsum += val;
  }
  bsum = fid_gsum<G>(bsum); wsum = fid_gsum<G>(wsum); fail = fid_gor<G>(fail);
  const int black = bsum / 28, white = wsum / 36;
  if (white - black < 40) fail = 1u;
  const int thr = (black + white) / 2;
  unsigned s_lo = 0, s_hi = 0, border_hi = 0;
#pragma unroll
  for (int m = 0; m < NS; ++m) {
    const int s = lg + G * m;
    if (s < 64) {
      const int r = s >> 3, c = s & 7;
      if (r == 0 || r == 7 || c == 0 || c == 7) { if (sv[m] >= thr) border_hi = 1u; }
      else if (sv[m] > thr) {
        const int b = 35 - ((r - 1) * 6 + (c - 1));
        if (b < 32) s_lo |= 1u << b; else s_hi |= 1u << (b - 32);
      }
    }
  }
  s_lo = fid_gor<G>(s_lo); s_hi = fid_gor<G>(s_hi); border_hi = fid_gor<G>(border_hi);
  if (border_hi) fail = 1u;
  uint64_t M = ((uint64_t)s_hi << 32) | s_lo;
  unsigned key = 0xFFFFFFFFu;                               // hamming * 4 ncodes + rotation * ncodes + id: the serial scan's order
  const unsigned n4 = 4u * (unsigned)ncodes;
  for (int rr = 0; rr < 4; ++rr) {
    for (int k = lg; k < ncodes; k += G) {
      const unsigned hd = (unsigned)__popcll(M ^ codes[k]);
      key = min(key, hd * n4 + (unsigned)(rr * ncodes + k));
    }
    M = fid_rot36(M);
  }
  key = fid_gmin<G>(key);
  if (fail || key == 0xFFFFFFFFu) return false;
  const unsigned hd = key / n4, rem = key - hd * n4;
  if ((int)hd > max_hamming) return false;
  ham = (int)hd; rot = (int)(rem / (unsigned)ncodes); id = (int)(rem - (unsigned)rot * (unsigned)ncodes);
  return true;
}

// ---- corner refinement of a quad in the refine_edges form (SURVEY.md appendix C.4 [U]: per edge >= 16 samples; at each, a
// scan along the normal in 0.25-px steps, weights (g2 - g1)^2 of the correct polarity, weighted-mean offset; a total-least-
// squares line through the refined points; corners = intersections of adjacent lines), restated [B] so that every decision
// is an integer and every floating-point step one rounded IEEE operation (+ - * / sqrt; -ffp-contract=off):
//   qi        the quad's corners rounded to pixels, clockwise on screen; black lies on the (-dy, dx) side of every edge a -> b
//   samples   16 per edge at a + alpha (b - a), alpha = (s + 2) / 19, s = 0..15 (the two positions nearest either corner are
//             left out: there the other edge bends the profile)
//   scan      17 bilinear samples (1/16-px fixed point, 256 x grey) along the OUTWARD normal n = (dy, -dx) / |d| at offsets
//             j / 2 px, j = -8 .. 8; step k = -6 .. 6 pairs j = k + 2 (one pixel further out) with j = k - 2 (further in):
//             weight (g1 - g2)^2 where g1 > g2 (white outside) and both samples lie inside the image, else 0;
//             offset = (sum k w / sum w) / 2 -- integer sums.  (Half-pixel steps: the same corner error against the
//             renderer's ground truth as C.4's quarter-pixel steps -- rms 0.039 px -- at half the loads, and the loads are
//             what this stage costs: scattered byte reads, 16 lanes per quad)
//   line      moments of the refined points relative to a, summed over the 16 samples along the pairing tree of a 16-lane
//             xor butterfly (offsets 8, 4, 2, 1; IEEE addition commutes, so every lane ends with the same totals);
//             centroid E, covariance C; the normal is the eigenvector of C's smaller eigenvalue in its well-conditioned
//             form; fewer than 4 valid samples or a vanishing normal: the line through the rounded corners
//   corners   corner c = intersection of the lines of edges c - 1 and c, solved relative to qi[c]; kept at qi[c] if the lines
//             are (nearly) parallel or the intersection lies more than 4 px away
// The 16 lanes of a decode group share one quad: lane s takes sample s of each of the four edges in turn; the line fit and the
// intersections are computed by every lane alike.  The CPU restatement the tests compare with follows the same steps, so
// the corners are the same bits.
__device__ __forceinline__ double fid_tree16(double v)
{
#ifdef RCC_FID_ABL_NOTREE
  return v * 16.0;                               // timing-only ablation: no cross-lane sums
#endif
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) v = v + __shfl_xor(v, off, 16);
  return v;
}
// 256 x grey at (px, py), position rounded to 1/16 px; the two taps of a row in one (unaligned) 16-bit load
__device__ __forceinline__ int fid_bil16(const uint8_t* __restrict__ g, int w, double px, double py)
{
  const int X = (int)rint(px * 16.0), Y = (int)rint(py * 16.0);
  const int ix = X >> 4, iy = Y >> 4, fx = X & 15, fy = Y & 15;
  const uint8_t* p = g + (size_t)iy * w + ix;
  unsigned short r0, r1;
  __builtin_memcpy(&r0, p, 2);
  __builtin_memcpy(&r1, p + w, 2);
  return (16 - fx) * (16 - fy) * (int)(r0 & 255) + fx * (16 - fy) * (int)(r0 >> 8) + (16 - fx) * fy * (int)(r1 & 255) + fx * fy * (int)(r1 >> 8);
}
__device__ void fid_refine_edges_group(const uint8_t* __restrict__ g, int w, int h, const int qi[8], int s /* lane of the group = sample */, bool valid, double qr[8])
{
  double E[4][2], V[4][2];
#pragma unroll 1
  for (int e = 0; e < 4; ++e) {
    const double ax = (double)qi[2 * e], ay = (double)qi[2 * e + 1], bx = (double)qi[2 * ((e + 1) & 3)], by = (double)qi[2 * ((e + 1) & 3) + 1];
    const double dx = bx - ax, dy = by - ay;
    const double dxx = dx * dx, dyy = dy * dy;
    const double L = sqrt(dxx + dyy);
    double nx = 0.0, ny = 0.0;
    if (L > 0.0) { nx = dy / L; ny = -dx / L; }
    const double alpha = (double)(s + 2) / 19.0;
    const double tx = alpha * dx, ty = alpha * dy;
    const double x0 = ax + tx, y0 = ay + ty;
    // 256 x grey at offset j / 2 along the normal (-1 outside the image), in three passes so that the 34 reads of an edge
    // are in flight together: positions and weights first, then every load, then the blends.  (Sample by sample, each
    // blend waited for its own two loads: 17 memory latencies per edge, 0.4 ms of this kernel per 1024 frames.)
    int P[17], off[17], fxy[17];
#pragma unroll
    for (int j = -8; j <= 8; ++j) {
      const double t = (double)j * 0.5;
      const double u = t * nx, v = t * ny;
      const double x = x0 + u, y = y0 + v;
      const bool in = valid && x >= 0.0 && y >= 0.0 && x <= (double)(w - 2) && y <= (double)(h - 2);
      const int X = (int)rint((in ? x : 0.0) * 16.0), Y = (int)rint((in ? y : 0.0) * 16.0);
      off[j + 8] = in ? (Y >> 4) * w + (X >> 4) : -1;
      fxy[j + 8] = (X & 15) | ((Y & 15) << 4);
    }
    unsigned short r0[17], r1[17];
#pragma unroll
    for (int j = 0; j < 17; ++j) {
#ifdef RCC_FID_ABL_NOLOAD
      r0[j] = (unsigned short)off[j]; r1[j] = (unsigned short)(off[j] >> 3);       // timing-only ablation: no image reads
#else
      const uint8_t* p = g + (size_t)(off[j] < 0 ? 0 : off[j]);
      __builtin_memcpy(&r0[j], p, 2);
      __builtin_memcpy(&r1[j], p + w, 2);
#endif
    }
#pragma unroll
    for (int j = 0; j < 17; ++j) {
      const int fx = fxy[j] & 15, fy = fxy[j] >> 4;
      const int b = (16 - fx) * (16 - fy) * (int)(r0[j] & 255) + fx * (16 - fy) * (int)(r0[j] >> 8) + (16 - fx) * fy * (int)(r1[j] & 255) + fx * fy * (int)(r1[j] >> 8);
      P[j] = off[j] < 0 ? -1 : b;
    }
    long long Mn = 0, Mc = 0;
#pragma unroll
    for (int k = -6; k <= 6; ++k) {
      const int g1 = P[k + 2 + 8], g2 = P[k - 2 + 8];
      if (g1 >= 0 && g2 >= 0 && g1 > g2) {
        const long long wt = (long long)(g1 - g2) * (long long)(g1 - g2);
        Mn += wt * k; Mc += wt;
      }
    }
    double rx = 0.0, ry = 0.0, one = 0.0;
    if (Mc != 0 && L > 0.0) {
      const double n0 = ((double)Mn / (double)Mc) * 0.5;
      const double ox = n0 * nx, oy = n0 * ny;
      rx = tx + ox; ry = ty + oy; one = 1.0;
    }
    const double N = fid_tree16(one), Sx = fid_tree16(rx), Sy = fid_tree16(ry);
    const double Sxx = fid_tree16(rx * rx), Sxy = fid_tree16(rx * ry), Syy = fid_tree16(ry * ry);
    E[e][0] = ax; E[e][1] = ay; V[e][0] = nx; V[e][1] = ny;
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
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int a = (c + 3) & 3, b = c;
    const double cx = (double)qi[2 * c], cy = (double)qi[2 * c + 1];
    const double ea = V[a][0] * (E[a][0] - cx) + V[a][1] * (E[a][1] - cy);
    const double eb = V[b][0] * (E[b][0] - cx) + V[b][1] * (E[b][1] - cy);
    const double det = V[a][0] * V[b][1] - V[a][1] * V[b][0];
    const double na = V[a][0] * V[a][0] + V[a][1] * V[a][1], nb = V[b][0] * V[b][0] + V[b][1] * V[b][1];
    double px = 0.0, py = 0.0;
    bool ok = det * det > 1e-6 * (na * nb);
    if (ok) {
      px = (ea * V[b][1] - eb * V[a][1]) / det;
      py = (V[a][0] * eb - V[b][0] * ea) / det;
      ok = (px * px + py * py <= 16.0);
    }
    qr[2 * c] = ok ? cx + px : cx;
    qr[2 * c + 1] = ok ? cy + py : cy;
  }
}

__global__ __launch_bounds__(256) void k_fid_quads(const uint8_t* __restrict__ grey, int w, int h, int min_contrast,
                                                   const rcc_cand* __restrict__ pre, const int32_t* __restrict__ npre,
                                                   const double* __restrict__ pre_xy, int kstride,
                                                   const uint64_t* __restrict__ codes, int ncodes, int max_hamming,
                                                   double tag_size, int max_targets, int refine_edges, double* __restrict__ ref_xy,
                                                   rcc_frame_corners* __restrict__ fc, rcc_detection* __restrict__ det,
                                                   int32_t* __restrict__ ndet)
{
  // per corner (index i of the kept list): successor, classification; per CLASSIFIED corner (compact index c, ascending
  // in i): packed rounded position, candidate row, i.  s_pk / s_cy are written by i in phase 1 and compacted in place.
  __shared__ __attribute__((aligned(16))) uint32_t s_pk[FID_MAXN];   // x | y << 16
  __shared__ int16_t s_cy[FID_MAXN];       // integer candidate row (the list stage's sort key): monotone, unlike the refined row
  __shared__ int16_t s_nxt[FID_MAXN];
  __shared__ int8_t s_dx[FID_MAXN], s_dy[FID_MAXN];
  __shared__ uint8_t s_ok[FID_MAXN], s_thr[FID_MAXN];
  __shared__ fid_hit s_hit[FID_MAXN];      // by the cycle's smallest corner index
  __shared__ int16_t s_cidx[FID_MAXN];     // i of compact corner c
  __shared__ int16_t s_quad[FID_MAXN / 4]; // smallest corner index of every 4-cycle (a corner has one successor: <= n / 4 cycles)
  __shared__ unsigned long long s_best[256];   // phase 2 (C): the lanes' best link so far, distance << 11 | compact index
  // (37.9 KB + 2 KB: four blocks per CU, so that 1024 frames are one round over 256 CUs)
  __shared__ int s_nc, s_nq;
  const int f = blockIdx.x, tid = threadIdx.x;
#ifdef RCC_FID_TRACE
  long long tk[12]; int tki = 0;
#define FID_TICK() do { __syncthreads(); tk[tki++] = wall_clock64(); } while (0)
#else
#define FID_TICK() do {} while (0)
#endif
  FID_TICK();
  rcc_frame_corners* out = fc + f;
  if (out->status != 0) { if (tid == 0) ndet[f] = 0; return; }
  const uint8_t* g = grey + (size_t)f * w * h;
  const double* xy = pre_xy + (size_t)f * kstride * 2;
  const int n = min(npre[f], FID_MAXN);
  // phase 1: corner classification at the rounded refined position
  for (int i = tid; i < n; i += 256) {
    const int x = (int)floor(xy[2 * i] + 0.5), y = (int)floor(xy[2 * i + 1] + 0.5);
    int dx = 0, dy = 0, t = 0;
    const bool ok = fid_corner_class(g, w, h, x, y, min_contrast, dx, dy, t);   // ok implies 5 <= x < w - 5 < 65536 (same for y)
    s_cy[i] = pre[(size_t)f * kstride + i].y;
    s_pk[i] = ((uint32_t)x & 0xFFFFu) | ((uint32_t)y << 16);
    s_ok[i] = ok; s_dx[i] = (int8_t)dx; s_dy[i] = (int8_t)dy; s_thr[i] = (uint8_t)t;
    s_nxt[i] = -1;
  }
  if (tid == 0) s_nq = 0;
  __syncthreads();
  FID_TICK();
  // ordered compaction of the classified corners: only they can be linked (typically a third of n), and ascending order
  // keeps the specification's tie-break (nearest, then smallest index).  In place: compact index <= i, and a chunk's 64
  // reads precede its writes (one wave, lockstep).
  if (tid < 64) {
    int base = 0;
    for (int c0 = 0; c0 < n; c0 += 64) {
      const int i = c0 + tid;
      const bool k = (i < n) && s_ok[i];
      const uint32_t pk = (i < n) ? s_pk[i] : 0u;
      const int16_t cy = (i < n) ? s_cy[i] : (int16_t)0;
      const unsigned long long bal = __ballot(k);
      const int c = base + __popcll(bal & ((1ull << tid) - 1ull));
      __builtin_amdgcn_wave_barrier();
      if (k) { s_cidx[c] = (int16_t)i; s_pk[c] = pk; s_cy[c] = cy; }
      base += __popcll(bal);
    }
    if (tid == 0) s_nc = base;
  }
  __syncthreads();
  const int nc = s_nc;
  FID_TICK();
  // phase 2: link along d1 (black on the (-dy, dx) side of the direction of travel).  The specification: among the corners
  // j that pass the integer gates (distance, direction, cone) AND whose connecting segment verifies (fid_edge_first + fid_edge_rest), the
  // nearest one, ties to the smallest index.  The verification reads 16 pixels; done inside the scan it ran whenever any
  // lane of the wave had a candidate (a memory latency per scan step).  So: (A) a pure-ALU scan over the packed compact
  // positions (one 16-byte LDS read per 4 corners) keeps the FK nearest gate-passers in (distance, index) order; (B) they
  // are probed together; (C) only if all FK fail and more passed the gates, the remaining passers are probed FK at a time.
  constexpr int FK = 4;
  for (int c0 = 0; c0 < nc; c0 += 256) {
    const int ci = c0 + tid;
    const bool act = ci < nc;
    const int i = act ? s_cidx[ci] : 0;
    const int dx = s_dx[i], dy = s_dy[i], dd = dx * dx + dy * dy;
    const uint32_t pki = s_pk[act ? ci : 0];
    const int xi = (int)(pki & 0xFFFFu), yi = (int)(pki >> 16), t = s_thr[i];
    // gates of compact corner c at packed position e: squared distance, or -1 (|cross| < 2^21: its square needs 64 bits)
    auto gate = [&](int c, uint32_t e, int& wx, int& wy) -> int {
      wx = (int)(e & 0xFFFFu) - xi; wy = (int)(e >> 16) - yi;
      const int ww = wx * wx + wy * wy;                      // <= 2 * 16384^2 (rcc_create limits the frame to 16384 x 16384)
      const int cr = wx * dy - wy * dx;
      const unsigned ac = (unsigned)abs(cr);
      const bool pass = (c != ci) && (ww >= 64) && (wx * dx + wy * dy > 0) &&
                        (8ull * ((unsigned long long)ac * ac) <= (unsigned long long)(unsigned)ww * (unsigned)dd);
      return pass ? ww : -1;
    };
    int cj[FK], cw[FK], npass = 0;
#pragma unroll
    for (int q = 0; q < FK; ++q) { cj[q] = -1; cw[q] = 0x7FFFFFFF; }
    // Half of the list is enough when the direction is steeper than the cone is wide (8 dy^2 > |d|^2: the cone's
    // half-angle has sin^2 = 1/8): every point of the cone then lies strictly on dy's side of this corner's row.  The
    // list is sorted by the integer candidate row, which differs from the refined, rounded row by at most 8
    // (k_subpix rejects moves beyond its window, <= 7), hence the slack of 16 rows.
    int lo = 0, hi = act ? nc : 0;
    if (act && 8 * dy * dy > dd) {
      const int yc = s_cy[ci];
      int a0 = 0, a1 = nc;
      if (dy > 0) {
        while (a0 < a1) { const int m = (a0 + a1) >> 1; if (s_cy[m] < yc - 16) a0 = m + 1; else a1 = m; }
        lo = a0;
      } else {
        while (a0 < a1) { const int m = (a0 + a1) >> 1; if (s_cy[m] <= yc + 16) a0 = m + 1; else a1 = m; }
        hi = a0;
      }
    }
    // (A)
    for (int c4 = lo & ~3; c4 < hi; c4 += 4) {
      const uint4 e4 = *reinterpret_cast<const uint4*>(&s_pk[c4]);
      const uint32_t ev[4] = { e4.x, e4.y, e4.z, e4.w };
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c4 + u;
        int wx, wy;
        const int ww = gate(c, ev[u], wx, wy);
        if (ww >= 0 && c >= lo && c < hi) {
          ++npass;
          // insert (ww, c) into the ascending list; c ascends along the scan, so an equal ww goes after its equals
          int jw = ww, jj = c;
#pragma unroll
          for (int q = 0; q < FK; ++q) {
            if (jw < cw[q]) { const int tw = cw[q], tj = cj[q]; cw[q] = jw; cj[q] = jj; jw = tw; jj = tj; }
          }
        }
      }
    }
    FID_TICK();     // (trace builds only; nc <= 256: the loop runs once)
    // (B) all FK probes are made unconditionally (an empty slot probes the corner against itself), so that the 16 FK loads
    // are in flight together; the first good one in (distance, index) order is the link
    bool okq[FK];
    int g1[FK], ewx[FK], ewy[FK];
#pragma unroll
    for (int q = 0; q < FK; ++q) {
      const uint32_t e = cj[q] >= 0 ? s_pk[cj[q]] : pki;
      ewx[q] = (int)(e & 0xFFFFu) - xi; ewy[q] = (int)(e >> 16) - yi;
      okq[q] = fid_edge_first(g, w, h, xi, yi, ewx[q], ewy[q], -dy, dx, t, g1[q]);
    }
#pragma unroll
    for (int q = 0; q < FK; ++q) {
      if (__any(okq[q])) {                               // wave-uniform: the other twelve loads only where some lane still needs them
        const bool r = fid_edge_rest(g, w, h, xi, yi, ewx[q], ewy[q], -dy, dx, t, g1[q]);
        okq[q] = okq[q] && r;
      }
    }
    int best = -1;
#pragma unroll
    for (int q = FK - 1; q >= 0; --q) if (act && cj[q] >= 0 && okq[q]) best = cj[q];
    FID_TICK();
    // (C) every remaining passer has to be probed (the nearest good one wins): corners of the code pattern pass the gates
    // with 25-100 others and link to none.  Probing inside the scan cost a memory latency per scan step for the whole wave;
    // probing FK at a time per lane left most lanes idle (the kernel is VALU-bound once enough frames are in flight).  So
    // the wave re-scans 8 corners per lane at a time, queues the passing (lane, corner) pairs in LDS, and probes the
    // queue with all 64 lanes; the nearest good corner per lane is an LDS atomic min of (distance, index).  A passer no
    // nearer than the lane's best so far is not queued.
    {
      const int lane = tid & 63, wbase = tid & ~63;
      uint32_t* const s_q = reinterpret_cast<uint32_t*>(s_hit) + (tid >> 6) * FID_QCAP;   // s_hit is idle until phase 3
      s_best[tid] = ~0ull;
      const bool need = act && best < 0 && npass > FK;
      int pos = need ? lo : hi;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      while (__any(pos < hi)) {
        const unsigned long long bk = s_best[tid];
        const int bestd = (bk == ~0ull) ? 0x7FFFFFFF : (int)(bk >> 11);
        int qn = 0;
        for (int st = 0; st < FID_QCAP / 64; ++st) {
          const bool in = pos < hi;
          const int c = in ? pos : 0;
          int wx, wy;
          const int ww = gate(c, s_pk[c], wx, wy);
          const bool pass = in && ww >= 0 && ww < bestd;
          const unsigned long long bal = __ballot(pass);
          if (pass) s_q[qn + __popcll(bal & ((1ull << lane) - 1ull))] = (uint32_t)lane | ((uint32_t)c << 6);
          qn += __popcll(bal);
          pos += in ? 1 : 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // first part of the segment test for every queued pair; the pairs that can still pass are compacted to the front of
        // the queue (in place: the write position never passes the read position, and a pass's 64 reads precede its writes)
        // with their count of right probes in bits 17..18
        int qm = 0;
        for (int p0 = 0; p0 < qn; p0 += 64) {
          const bool in = p0 + lane < qn;
          const uint32_t pr = in ? s_q[p0 + lane] : 0u;
          int good1 = 0;
          bool more = false;
          if (in) {      // lanes beyond the queue's end take no part: slot 0 of s_cidx / s_pk may never have been written (no cross-lane operation inside)
            const int L = (int)(pr & 63u), c = (int)(pr >> 6);
            const int cL = c0 + wbase + L, iL = s_cidx[cL];                    // lane L of this wave is active: it queued the pair
            const uint32_t pkL = s_pk[cL], e = s_pk[c];
            const int xL = (int)(pkL & 0xFFFFu), yL = (int)(pkL >> 16);
            more = fid_edge_first(g, w, h, xL, yL, (int)(e & 0xFFFFu) - xL, (int)(e >> 16) - yL, -(int)s_dy[iL], (int)s_dx[iL], (int)s_thr[iL], good1);
          }
          const unsigned long long bal = __ballot(more);
          __builtin_amdgcn_wave_barrier();
          if (more) s_q[qm + __popcll(bal & ((1ull << lane) - 1ull))] = pr | ((uint32_t)good1 << 17);
          qm += __popcll(bal);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int p = lane; p < qm; p += 64) {
          const uint32_t pr = s_q[p];
          const int L = (int)(pr & 63u), c = (int)((pr >> 6) & 2047u), good1 = (int)(pr >> 17);
          const int cL = c0 + wbase + L, iL = s_cidx[cL];
          const uint32_t pkL = s_pk[cL], e = s_pk[c];
          const int xL = (int)(pkL & 0xFFFFu), yL = (int)(pkL >> 16);
          const int dxL = s_dx[iL], dyL = s_dy[iL], tL = s_thr[iL];
          const int wx = (int)(e & 0xFFFFu) - xL, wy = (int)(e >> 16) - yL;
          if (fid_edge_rest(g, w, h, xL, yL, wx, wy, -dyL, dxL, tL, good1))
            atomicMin(&s_best[wbase + L], ((unsigned long long)(unsigned)(wx * wx + wy * wy) << 11) | (unsigned long long)c);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
      const unsigned long long bk = s_best[tid];
      if (need && bk != ~0ull) best = (int)(bk & 2047ull);
    }
    if (act && best >= 0) s_nxt[i] = s_cidx[best];
  }
  __syncthreads();
  FID_TICK();
  // phase 3: 4-cycles, listed by their smallest index
  for (int i = tid; i < n; i += 256) {
    s_hit[i].id = -1;                      // (the array was the link phase's pair queue)
    if (!s_ok[i]) continue;
    const int j = s_nxt[i]; if (j < 0) continue;
    const int k = s_nxt[j]; if (k < 0) continue;
    const int l = s_nxt[k]; if (l < 0) continue;
    if (s_nxt[l] != i) continue;
    if (j == k || j == l || k == l || k == i || j == i || l == i) continue;
    if (!(i < j && i < k && i < l)) continue;
    s_quad[atomicAdd(&s_nq, 1)] = (int16_t)i;
  }
  __syncthreads();
  FID_TICK();
  // decode: FID_G lanes per quad
  {
    constexpr int G = 16, NG = 256 / G;
    const int nq = s_nq, gid = tid / G, lg = tid % G;
    for (int qb = 0; qb < nq; qb += NG) {
      const bool have = qb + gid < nq;
      const int i = have ? s_quad[qb + gid] : 0;
      int idx[4];
      idx[0] = i;
      for (int c = 1; c < 4; ++c) { const int nx = have ? s_nxt[idx[c - 1]] : 0; idx[c] = nx; }
      double q[8];
      for (int c = 0; c < 4; ++c) { q[2 * c] = have ? xy[2 * idx[c]] : 0.0; q[2 * c + 1] = have ? xy[2 * idx[c] + 1] : 0.0; }
      bool go = have;
      if (refine_edges) {
        // refine_edges form: the quad's corners come from its edges, started at the rounded positions the quad was found on
        int qi[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) qi[c] = (int)floor(q[c] + 0.5);
        const long long cri = (long long)(qi[2] - qi[0]) * (qi[5] - qi[3]) - (long long)(qi[3] - qi[1]) * (qi[4] - qi[2]);
        go = have && (cri > 0);
#ifndef RCC_FID_ABL_NOREFINE
        fid_refine_edges_group(g, w, h, qi, lg, go, q);
#endif
        if (go && lg == 0) {
          double* r = ref_xy + (size_t)f * kstride * 2;
#pragma unroll
          for (int c = 0; c < 4; ++c) { r[2 * idx[c]] = q[2 * c]; r[2 * idx[c] + 1] = q[2 * c + 1]; }
        }
      }
      const double e1 = (q[2] - q[0]) * (q[5] - q[3]), e2 = (q[3] - q[1]) * (q[4] - q[2]);
      const double cr = e1 - e2;
      int id = 0, ham = 0, rot = 0;
      const bool hit = fid_decode_group<G>(g, w, h, q, codes, ncodes, max_hamming, lg, go && (cr > 0.0), id, ham, rot);
      if (hit && lg == 0) { fid_hit hrec; hrec.id = (int16_t)id; hrec.ham = (int8_t)ham; hrec.rot = (int8_t)rot; s_hit[i] = hrec; }
    }
  }
  __threadfence_block();                   // the refined corners (global) are read back by this block's emission below
  __syncthreads();
  FID_TICK();
  // ordered emission (by the cycle's smallest index, as the specification's scan does): wave 0, one lane per hit
  if (tid < 64) {
    int base = 0;
    for (int c0 = 0; c0 < n; c0 += 64) {
      const int i = c0 + tid;
      const bool hit = (i < n) && s_hit[i].id >= 0;
      const unsigned long long bal = __ballot(hit);
      const int m = base + __popcll(bal & ((1ull << tid) - 1ull));
      if (hit && m < max_targets) {
        const fid_hit hrec = s_hit[i];
        rcc_detection d;
        d.frame = f; d.id = hrec.id; d.hamming = hrec.ham; d.ncorners = 4; d.size = tag_size;
        int idx[4];
        idx[0] = i; idx[1] = s_nxt[i]; idx[2] = s_nxt[idx[1]]; idx[3] = s_nxt[idx[2]];
        const int rot = hrec.rot;
        const int ord[4] = { (rot + 3) & 3, (rot + 2) & 3, (rot + 1) & 3, rot & 3 };   // bl, br, tr, tl
        const double* cxy = refine_edges ? ref_xy + (size_t)f * kstride * 2 : xy;     // (written by this block's decode stage, before the barrier)
        for (int c = 0; c < 4; ++c) { d.corners[c][0] = cxy[2 * idx[ord[c]]]; d.corners[c][1] = cxy[2 * idx[ord[c]] + 1]; }
        for (int c = 0; c < 3; ++c) { d.rvec[c] = 0.0; d.tvec[c] = 0.0; }
        d.rms = 0.0; d.pnp_status = 0; d.pnp_iters = 0;
        det[(size_t)f * max_targets + m] = d;
      }
      base += __popcll(bal);
    }
    if (tid == 0) {
      const int kept = min(base, max_targets);
      ndet[f] = kept;
      out->nkept = n;
      out->ncorners = 0;
      if (kept == 0) out->status |= RCC_FRAME_NOT_FOUND;
    }
  }
#ifdef RCC_FID_TRACE
  FID_TICK();
  if (tid == 0 && (f & 63) == 0)
    printf("fid f=%d n=%d nc=%d  class %lld  compact %lld  link: scan %lld nearest-4 probes %lld the rest %lld  cycles %lld  decode %lld  emit %lld (x10ns)\n", f, n, nc,
           tk[1] - tk[0], tk[2] - tk[1], tk[3] - tk[2], tk[4] - tk[3], tk[5] - tk[4], tk[6] - tk[5], tk[7] - tk[6], tk[8] - tk[7]);
#endif
}

// a7 for the tags: one lane per detection (camera_pose.cpp:152-163: four corners bl,br,tr,tl,
// object points (+-s/2, +-s/2, 0))
__global__ __launch_bounds__(64) void k_pnp_tags(rcc_detection* __restrict__ det, const int32_t* __restrict__ ndet,
                                                 int nframes, int max_targets, double tag_size, int reference_mode, rcc_cam cam)
{
  const int t = blockIdx.x * 64 + threadIdx.x;
  if (t >= nframes * max_targets) return;
  const int f = t / max_targets, k = t - f * max_targets;
  if (k >= ndet[f]) return;
  rcc_detection* d = det + t;
  const double s2 = 0.5 * tag_size;
  double obj[12] = { -s2, -s2, 0, s2, -s2, 0, s2, s2, 0, -s2, s2, 0 };
  double img[8];
  for (int c = 0; c < 4; ++c) {
    double x = d->corners[c][0], y = d->corners[c][1];
    if (reference_mode) { x = (double)(int)x; y = (double)(int)y; }       // corner_detections.cpp:53-54
    img[2 * c] = x; img[2 * c + 1] = y;
  }
  rccpnp::Pts p{ obj, img, 4 };
  rccpnp::Cam cm;
  cm.fx = cam.fx; cm.fy = cam.fy; cm.cx = cam.cx; cm.cy = cam.cy;
  for (int i = 0; i < 5; ++i) cm.k[i] = cam.D[i];
  cm.solver = cam.solver;
  double r[3], tv[3], e = 0.0;
  double wsl[rccpnp::PNP_WS];
  int it = 0;
  const int st = rccpnp::solve_pnp(rccpnp::SerialPar{ wsl }, p, cm, cam.model, r, tv, &e, &it);
  for (int c = 0; c < 3; ++c) { d->rvec[c] = r[c]; d->tvec[c] = tv[c]; }
  d->rms = e; d->pnp_status = st; d->pnp_iters = it;
}

hipError_t rcc_launch_fid(rcc_handle* h, const uint8_t* d_grey, int nframes, hipStream_t s)
{
  const rcc_config& c = h->cfg;
  if (nframes <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_fid_quads, dim3(nframes), dim3(256), 0, s, d_grey, c.width, c.height, c.thr_min_contrast,
                     h->d_pre, h->d_npre, h->d_pre_xy, h->kept_cap, h->d_family, c.family_n, c.tag_max_hamming,
                     c.tag_size, c.max_targets, c.tag_refine == RCC_TAG_REFINE_EDGES ? 1 : 0, h->d_ref_xy, h->d_fc, h->d_det, h->d_ndet);
  return hipGetLastError();
}

hipError_t rcc_launch_pnp_tags(rcc_handle* h, int nframes, hipStream_t s)
{
  const rcc_config& c = h->cfg;
  if (nframes <= 0) return hipSuccess;
  rcc_cam cam;
  cam.fx = c.K[0]; cam.cx = c.K[2]; cam.fy = c.K[4]; cam.cy = c.K[5];
  for (int i = 0; i < 8; ++i) cam.D[i] = c.D[i];
  cam.model = h->undist ? RCC_DIST_NONE : c.dist_model;
  cam.solver = h->pnp_solver;
  const int total = nframes * c.max_targets;
  if (h->pnp_use_mfma) return rcc_launch_pnp_tags_mfma(h, nframes, cam, s);   // J^T J / J^T e on the matrix cores (k_pnp_mfma.hip)
  hipLaunchKernelGGL(k_pnp_tags, dim3((total + 63) / 64), dim3(64), 0, s, h->d_det, h->d_ndet, nframes, c.max_targets,
                     c.tag_size, c.reference_mode, cam);
  return hipGetLastError();
}
