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
// One 256-thread block per frame: the work is O(n^2) pair gating over n <= 2048 refined corners
// plus a handful of decodes -- latency-bound, never bandwidth-bound; frames are the parallel axis.
#include "rcc_internal.h"
#define RCC_PNP_NOINLINE 1
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

// all 16 probes are loaded before any is tested: one memory latency instead of up to eight dependent ones
// (the decision is unchanged: false if any probe leaves the image, else at least 7 of 8 two-sided probes good)
__device__ bool fid_edge_ok(const uint8_t* __restrict__ g, int w, int h, int xi, int yi, int wx, int wy, int nx, int ny, int t)
{
  const int ox = fid_rdiv10(nx), oy = fid_rdiv10(ny);
  bool inb = true;
  int vb[8], vc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int mx = (xi * 16 + wx * (2 * k + 1) + 8) >> 4, my = (yi * 16 + wy * (2 * k + 1) + 8) >> 4;
    const int bx = mx + ox, by = my + oy, cx = mx - ox, cy = my - oy;
    const bool ok = !(bx < 0 || by < 0 || bx >= w || by >= h || cx < 0 || cy < 0 || cx >= w || cy >= h);
    inb = inb && ok;
    const int bxc = min(max(bx, 0), w - 1), byc = min(max(by, 0), h - 1), cxc = min(max(cx, 0), w - 1), cyc = min(max(cy, 0), h - 1);
    vb[k] = g[(size_t)byc * w + bxc];
    vc[k] = g[(size_t)cyc * w + cxc];
  }
  int good = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) good += (vb[k] <= t && vc[k] > t) ? 1 : 0;
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

__device__ bool fid_decode(const uint8_t* __restrict__ g, int w, int h, const double q[8], const uint64_t* __restrict__ codes,
                           int ncodes, int max_hamming, int& id, int& ham, int& rot)
{
  double H[9];
  if (!fid_homography(q, H)) return false;
  uint64_t S = 0, border_hi = 0;      // payload bits are decided after the levels are known: keep the samples
  uint8_t cell[64];
  int bsum = 0, wsum = 0;
  for (int r = 0; r < 8; ++r)
    for (int c = 0; c < 8; ++c) {
      const int s = fid_sample(g, w, h, H, (double)c + 0.5, (double)r + 0.5);
      if (s < 0) return false;
      cell[r * 8 + c] = (uint8_t)s;
      if (r == 0 || r == 7 || c == 0 || c == 7) bsum += s;
    }
  for (int i = -1; i <= 8; ++i) {
    const int s0 = fid_sample(g, w, h, H, (double)i + 0.5, -0.5), s1 = fid_sample(g, w, h, H, (double)i + 0.5, 8.5);
    if (s0 < 0 || s1 < 0) return false;
    wsum += s0 + s1;
    if (i >= 0 && i <= 7) {
      const int s2 = fid_sample(g, w, h, H, -0.5, (double)i + 0.5), s3 = fid_sample(g, w, h, H, 8.5, (double)i + 0.5);
      if (s2 < 0 || s3 < 0) return false;
      wsum += s2 + s3;
    }
  }
  const int black = bsum / 28, white = wsum / 36;
  if (white - black < 40) return false;
  const int thr = (black + white) / 2;
  for (int r = 0; r < 8; ++r)
    for (int c = 0; c < 8; ++c) {
      const int s = cell[r * 8 + c];
      if (r == 0 || r == 7 || c == 0 || c == 7) { if (s >= thr) border_hi = 1; }
      else if (s > thr) S |= (uint64_t)1 << (35 - ((r - 1) * 6 + (c - 1)));
    }
  if (border_hi) return false;
  int best_id = -1, best_h = 99, best_rot = 0;
  uint64_t M = S;
  for (int rr = 0; rr < 4; ++rr) {
    for (int k = 0; k < ncodes; ++k) {
      const int hd = __popcll(M ^ codes[k]);
      if (hd < best_h) { best_h = hd; best_id = k; best_rot = rr; }
    }
    M = fid_rot36(M);
  }
  if (best_id < 0 || best_h > max_hamming) return false;
  id = best_id; ham = best_h; rot = best_rot;
  return true;
}

struct fid_hit { int16_t j, k, l, pad; int32_t id; int16_t ham, rot; };

__global__ __launch_bounds__(256) void k_fid_quads(const uint8_t* __restrict__ grey, int w, int h, int min_contrast,
                                                   const rcc_cand* __restrict__ pre, const int32_t* __restrict__ npre,
                                                   const double* __restrict__ pre_xy, int kstride,
                                                   const uint64_t* __restrict__ codes, int ncodes, int max_hamming,
                                                   double tag_size, int max_targets,
                                                   rcc_frame_corners* __restrict__ fc, rcc_detection* __restrict__ det,
                                                   int32_t* __restrict__ ndet)
{
  __shared__ int16_t s_px[FID_MAXN], s_py[FID_MAXN], s_nxt[FID_MAXN];
  __shared__ int16_t s_cy[FID_MAXN];       // integer candidate row (the list stage's sort key): monotone, unlike the refined row
  __shared__ int8_t s_dx[FID_MAXN], s_dy[FID_MAXN];
  __shared__ uint8_t s_ok[FID_MAXN], s_thr[FID_MAXN], s_hitf[FID_MAXN];
  __shared__ fid_hit s_hit[FID_MAXN];
  __shared__ int16_t s_cidx[FID_MAXN];     // indices of the convex black corners, ascending
  __shared__ int s_nc;
  const int f = blockIdx.x, tid = threadIdx.x;
  rcc_frame_corners* out = fc + f;
  if (out->status != 0) { if (tid == 0) ndet[f] = 0; return; }
  const uint8_t* g = grey + (size_t)f * w * h;
  const double* xy = pre_xy + (size_t)f * kstride * 2;
  const int n = min(npre[f], FID_MAXN);
  // phase 1: corner classification at the rounded refined position
  for (int i = tid; i < n; i += 256) {
    const int x = (int)floor(xy[2 * i] + 0.5), y = (int)floor(xy[2 * i + 1] + 0.5);
    int dx = 0, dy = 0, t = 0;
    const bool ok = fid_corner_class(g, w, h, x, y, min_contrast, dx, dy, t);
    s_cy[i] = pre[(size_t)f * kstride + i].y;
    s_px[i] = (int16_t)x; s_py[i] = (int16_t)y; s_ok[i] = ok; s_dx[i] = (int8_t)dx; s_dy[i] = (int8_t)dy; s_thr[i] = (uint8_t)t;
    s_hitf[i] = 0;
  }
  __syncthreads();
  // ordered compaction of the classified corners: only they can be linked (typically a third of n),
  // and ascending order keeps the specification's tie-break (nearest, then smallest index)
  if (tid < 64) {
    int base = 0;
    for (int c0 = 0; c0 < n; c0 += 64) {
      const int i = c0 + tid;
      const bool k = (i < n) && s_ok[i];
      const unsigned long long bal = __ballot(k);
      if (k) s_cidx[base + __popcll(bal & ((1ull << tid) - 1ull))] = (int16_t)i;
      base += __popcll(bal);
    }
    if (tid == 0) s_nc = base;
  }
  __syncthreads();
  const int nc = s_nc;
  // phase 2: link along d1 (black on the (-dy, dx) side of the direction of travel): nearest accepted
  for (int i = tid; i < n; i += 256) s_nxt[i] = -1;
  __syncthreads();
  // The specification: among the corners j that pass the integer gates (distance, direction, cone) AND whose
  // connecting segment verifies (fid_edge_ok), the nearest one, ties to the smallest index.  The verification
  // reads 16 pixels; done inside the scan it ran whenever any lane of the wave had a candidate (a few memory
  // latencies per scan step: 4 ms per batch).  So: (A) a pure-ALU scan keeps the FK nearest gate-passers in
  // (distance, index) order; (B) they are verified in that order, all lanes in step; (C) only if all FK fail and
  // more passed the gates, the plain scan finishes the job (rare; keeps the result exact).
  constexpr int FK = 4;
  for (int c0 = 0; c0 < nc; c0 += 256) {
    const int ci = c0 + tid;
    const bool act = ci < nc;
    const int i = act ? s_cidx[ci] : 0;
    const int dx = s_dx[i], dy = s_dy[i], dd = dx * dx + dy * dy;
    const int xi = s_px[i], yi = s_py[i], t = s_thr[i];
    int cj[FK], cw[FK], npass = 0;
#pragma unroll
    for (int q = 0; q < FK; ++q) { cj[q] = -1; cw[q] = 0x7FFFFFFF; }
    // Half of the list is enough when the direction is steeper than the cone is wide (8 dy^2 > |d|^2: the cone's
    // half-angle has sin^2 = 1/8): every point of the cone then lies strictly on dy's side of this corner's row.  The
    // list is sorted by the integer candidate row, which differs from the refined, rounded row by at most 8
    // (k_subpix rejects moves beyond its window, <= 7), hence the slack of 16 rows.
    int lo = 0, hi = nc;
    if (act && 8 * dy * dy > dd) {
      const int yc = s_cy[i];
      int a0 = 0, a1 = nc;
      if (dy > 0) {
        while (a0 < a1) { const int m = (a0 + a1) >> 1; if (s_cy[s_cidx[m]] < yc - 16) a0 = m + 1; else a1 = m; }
        lo = a0;
      } else {
        while (a0 < a1) { const int m = (a0 + a1) >> 1; if (s_cy[s_cidx[m]] <= yc + 16) a0 = m + 1; else a1 = m; }
        hi = a0;
      }
    }
    if (act) {
      for (int cjx = lo; cjx < hi; ++cjx) {
        const int j = s_cidx[cjx];
        const int wx = s_px[j] - xi, wy = s_py[j] - yi;
        const int ww = wx * wx + wy * wy;                    // <= 2 * 16384^2 fits int32
        const long long cr = (long long)wx * dy - (long long)wy * dx;
        const bool pass = (j != i) && (ww >= 64) && (wx * dx + wy * dy > 0) && (8 * cr * cr <= (long long)ww * dd);
        if (pass) {
          ++npass;
          // insert (ww, j) into the ascending list; j ascends along the scan, so an equal ww goes after its equals
          int jw = ww, jj = j;
#pragma unroll
          for (int q = 0; q < FK; ++q) {
            if (jw < cw[q]) { const int tw = cw[q], tj = cj[q]; cw[q] = jw; cj[q] = jj; jw = tw; jj = tj; }
          }
        }
      }
    }
    int best = -1;
#pragma unroll
    for (int q = 0; q < FK; ++q) {
      if (act && best < 0 && cj[q] >= 0) {
        const int j = cj[q];
        if (fid_edge_ok(g, w, h, xi, yi, s_px[j] - xi, s_py[j] - yi, -dy, dx, t)) best = j;
      }
    }
    if (act && best < 0 && npass > FK) {
      int bestd = 0;
      for (int cjx = lo; cjx < hi; ++cjx) {
        const int j = s_cidx[cjx];
        if (j == i) continue;
        const int wx = s_px[j] - xi, wy = s_py[j] - yi;
        const int ww = wx * wx + wy * wy;
        if (ww < 64) continue;
        if (wx * dx + wy * dy <= 0) continue;
        const long long cr = (long long)wx * dy - (long long)wy * dx;
        if (8 * cr * cr > (long long)ww * dd) continue;
        if (best >= 0 && ww >= bestd) continue;
        if (!fid_edge_ok(g, w, h, xi, yi, wx, wy, -dy, dx, t)) continue;
        best = j; bestd = ww;
      }
    }
    if (act) s_nxt[i] = (int16_t)best;
  }
  __syncthreads();
  // phase 3: 4-cycles from their smallest index, decode
  for (int i = tid; i < n; i += 256) {
    if (!s_ok[i]) continue;
    const int j = s_nxt[i]; if (j < 0) continue;
    const int k = s_nxt[j]; if (k < 0) continue;
    const int l = s_nxt[k]; if (l < 0) continue;
    if (s_nxt[l] != i) continue;
    if (j == k || j == l || k == l || k == i || j == i || l == i) continue;
    if (!(i < j && i < k && i < l)) continue;
    const int idx[4] = { i, j, k, l };
    double q[8];
    for (int c = 0; c < 4; ++c) { q[2 * c] = xy[2 * idx[c]]; q[2 * c + 1] = xy[2 * idx[c] + 1]; }
    const double e1 = (q[2] - q[0]) * (q[5] - q[3]), e2 = (q[3] - q[1]) * (q[4] - q[2]);
    const double cr = e1 - e2;
    if (!(cr > 0.0)) continue;
    int id, ham, rot;
    if (!fid_decode(g, w, h, q, codes, ncodes, max_hamming, id, ham, rot)) continue;
    fid_hit hrec;
    hrec.j = (int16_t)j; hrec.k = (int16_t)k; hrec.l = (int16_t)l; hrec.pad = 0; hrec.id = id; hrec.ham = (int16_t)ham; hrec.rot = (int16_t)rot;
    s_hit[i] = hrec;
    s_hitf[i] = 1;
  }
  __syncthreads();
  // ordered emission (by the cycle's smallest index, as the specification's scan does)
  if (tid == 0) {
    int m = 0;
    for (int i = 0; i < n; ++i) {
      if (!s_hitf[i]) continue;
      if (m < max_targets) {
        const fid_hit hrec = s_hit[i];
        rcc_detection d;
        d.frame = f; d.id = hrec.id; d.hamming = hrec.ham; d.ncorners = 4; d.size = tag_size;
        const int idx[4] = { i, hrec.j, hrec.k, hrec.l };
        const int rot = hrec.rot;
        const int ord[4] = { (rot + 3) & 3, (rot + 2) & 3, (rot + 1) & 3, rot & 3 };   // bl, br, tr, tl
        for (int c = 0; c < 4; ++c) { d.corners[c][0] = xy[2 * idx[ord[c]]]; d.corners[c][1] = xy[2 * idx[ord[c]] + 1]; }
        for (int c = 0; c < 3; ++c) { d.rvec[c] = 0.0; d.tvec[c] = 0.0; }
        d.rms = 0.0; d.pnp_status = 0; d.pnp_iters = 0;
        det[(size_t)f * max_targets + m] = d;
      }
      ++m;
    }
    const int kept = min(m, max_targets);
    ndet[f] = kept;
    out->nkept = n;
    out->ncorners = 0;
    if (kept == 0) out->status |= RCC_FRAME_NOT_FOUND;
  }
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
                     c.tag_size, c.max_targets, h->d_fc, h->d_det, h->d_ndet);
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
  hipLaunchKernelGGL(k_pnp_tags, dim3((total + 63) / 64), dim3(64), 0, s, h->d_det, h->d_ndet, nframes, c.max_targets,
                     c.tag_size, c.reference_mode, cam);
  return hipGetLastError();
}
