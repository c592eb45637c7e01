// grid_frame.h -- the per-frame body of stages a4.3 + a6 (see k_grid.hip) as a device function, so that it can
// run as its own kernel (k_validate_grid: stage API, any caller-supplied binary image) or in front of the board
// pose solve in one kernel (k_grid_pnp in k_pnp.hip: both are one-wavefront-per-frame dependency chains, and a
// frame's pose needs only that frame's lattice).
#pragma once
#include "rcc_internal.h"
#include "wave_reduce.h"

#define GM 24
#define GW (2 * GM + 1)
#define GBOARD 16

__constant__ int8_t c_ring16[16][2] = {
  { 5, 0}, { 5, 2}, { 4, 4}, { 2, 5}, { 0, 5}, {-2, 5}, {-4, 4}, {-5, 2},
  {-5, 0}, {-5,-2}, {-4,-4}, {-2,-5}, { 0,-5}, { 2,-5}, { 4,-4}, { 5,-2}
};

// whole-wave min / max / sum: exchanges by permlane swaps and DPP (wave_reduce.h), no ds_bpermute
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
  return wred::all_reduce64(v, [](unsigned long long a, unsigned long long b) { return a < b ? a : b; });
}
__device__ __forceinline__ int wave_min_i32(int v)
{
  return (int)wred::all_reduce32((unsigned)v, [](unsigned a, unsigned b) { return (unsigned)min((int)a, (int)b); });
}
__device__ __forceinline__ int wave_max_i32(int v)
{
  return (int)wred::all_reduce32((unsigned)v, [](unsigned a, unsigned b) { return (unsigned)max((int)a, (int)b); });
}
__device__ __forceinline__ long long wave_sum_i64(long long v)
{
  return (long long)wred::all_reduce64((unsigned long long)v, [](unsigned long long a, unsigned long long b) { return a + b; });
}

// binary-image value at (x, y): from the full image, or (thr != null) from the grey image and the compact
// threshold map the band kernel wrote for rcc_detect_batch -- the same value by definition (a3)
struct BinSrc {
  const uint8_t* bin;     // frame's binary image, or null
  const uint8_t* grey;    // frame's grey image
  const uint8_t* thr;     // frame's compact map [band][tile row][RCC_THR_PITCH], or null
  int w, th;
  __device__ __forceinline__ int at(int x, int y) const
  {
    if (!thr) return bin[(size_t)y * w + x];
    const int band = x / RCC_BAND_W;
    const int lv = thr[((size_t)band * th + (y >> 2)) * RCC_THR_PITCH + ((x - band * RCC_BAND_W) >> 2)];
    if (lv == 255) return 127;
    return grey[(size_t)y * w + x] > lv ? 255 : 0;
  }
};

// The 16 ring samples are fetched in two rounds of independent loads -- all tile levels, then all grey values (always
// in-image addresses; the grey value is simply not used where the level says "flat") -- instead of 16 dependent
// level -> grey pairs one after the other: the ring test was 60 us of the 260 us this one-wave-per-frame kernel takes.
__device__ __forceinline__ bool ring_ok(const BinSrc& b, int w, int h, int x, int y)
{
  if (x < 5 || y < 5 || x >= w - 5 || y >= h - 5) return false;
  int v[16];
  if (!b.thr) {
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = b.bin[(size_t)(y + c_ring16[k][1]) * w + (x + c_ring16[k][0])];
  } else {
    int lv[16], g[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int xx = x + c_ring16[k][0], yy = y + c_ring16[k][1];
      const int band = xx / RCC_BAND_W;
      lv[k] = b.thr[((size_t)band * b.th + (yy >> 2)) * RCC_THR_PITCH + ((xx - band * RCC_BAND_W) >> 2)];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) g[k] = b.grey[(size_t)(y + c_ring16[k][1]) * w + (x + c_ring16[k][0])];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = (lv[k] == 255) ? 127 : (g[k] > lv[k] ? 255 : 0);
  }
  int tr = 0;
  bool any127 = false;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    any127 |= (v[k] == 127);
    tr += (v[k] != v[(k + 1) & 15]);
  }
  return !any127 && tr == 4;
}

struct grid_smem {
  int32_t px[RCC_MAX_KEPT], py[RCC_MAX_KEPT];
  int32_t score[RCC_MAX_KEPT];
  double xy[2 * RCC_MAX_KEPT];
  int16_t rx[RCC_MAX_KEPT], ry[RCC_MAX_KEPT];
  uint8_t ok[RCC_MAX_KEPT], keep[RCC_MAX_KEPT], used[RCC_MAX_KEPT], taken[RCC_MAX_KEPT];
  int16_t lab[GW * GW];
  int16_t qi[RCC_MAX_KEPT], qj[RCC_MAX_KEPT];
  int16_t tmp[RCC_MAX_KEPT], t2[RCC_MAX_KEPT];
  int32_t order[RCC_MAX_KEPT];
};

#define LAB(i, j) sm.lab[((i) + GM) * GW + ((j) + GM)]

// nearest point with used[k]==0 to (qx,qy); ties -> smaller index; returns -1 if none.
// SMALL (w^2 + h^2 < 2^24: every distance between two image points fits 24 bits): the key (distance << 8 | index) is one
// dword and the wave-wide minimum a 32-bit reduction -- half the instructions of the 64-bit form, in a loop that runs ~200
// times per frame on a single wave.  A query outside the image may be farther than 2^24 - 2 from every point; its distance
// saturates there, and such a match is refused by the caller either way (8 * distance > step^2, step^2 < 2^24).
template <bool SMALL>
__device__ __forceinline__ int nearest_free(const grid_smem& sm, int n, int lane, long long qx, long long qy, long long* dist)
{
  if (SMALL) {
    const int ix = (int)qx, iy = (int)qy;
    unsigned best = ~0u;
    for (int k = lane; k < n; k += 64) {
      if (sm.used[k]) continue;
      const int dx = sm.px[k] - ix, dy = sm.py[k] - iy;
      const unsigned d = min((unsigned)(dx * dx) + (unsigned)(dy * dy), 0xFFFFFEu);
      const unsigned key = (d << 8) | (unsigned)k;
      best = key < best ? key : best;
    }
    best = wred::all_reduce32(best, [](unsigned a, unsigned b) { return a < b ? a : b; });
    if (best == ~0u) return -1;
    *dist = (long long)(best >> 8);
    return (int)(best & 255u);
  }
  unsigned long long best = ~0ull;
  for (int k = lane; k < n; k += 64) {
    if (sm.used[k]) continue;
    long long dx = (long long)sm.px[k] - qx, dy = (long long)sm.py[k] - qy;
    unsigned long long key = ((unsigned long long)(dx * dx + dy * dy) << 8) | (unsigned long long)k;
    best = key < best ? key : best;
  }
  best = wave_min_u64(best);
  if (best == ~0ull) return -1;
  *dist = (long long)(best >> 8);
  return (int)(best & 255ull);
}

// returns true when the board lattice was found: its corners are then also at sm.xy[2 * sm.order[k]], k = 0..cols*rows-1
__device__ __forceinline__ bool grid_frame(grid_smem& sm, const int f, const int lane,
                                           const uint8_t* __restrict__ bin, const uint8_t* __restrict__ grey,
                                           const uint8_t* __restrict__ thr, int nbands, int w, int h,
                                           const rcc_cand* __restrict__ pre, const int32_t* __restrict__ npre,
                                           const double* __restrict__ pre_xy, int xj_check, int dedupe_radius,
                                           int target_kind, int cols, int rows,
                                           rcc_frame_corners* __restrict__ fc,
                                           rcc_cand* __restrict__ kept_out, double* __restrict__ kept_xy_out)
{
  rcc_frame_corners* out = fc + f;
  if (out->status != 0) return false;   // overflow flagged by the list stage: the frame yields nothing
  BinSrc b;
  b.bin = bin ? bin + (size_t)f * w * h : nullptr;
  b.grey = grey + (size_t)f * w * h;
  b.thr = thr ? thr + (size_t)f * nbands * (h >> 2) * RCC_THR_PITCH : nullptr;
  b.w = w; b.th = h >> 2;
  const int n = npre[f];

  // ---- a4.3 validation at the rounded refined position
  for (int i = lane; i < n; i += 64) {
    const double x = pre_xy[((size_t)f * RCC_MAX_KEPT + i) * 2], y = pre_xy[((size_t)f * RCC_MAX_KEPT + i) * 2 + 1];
    int xi = (int)floor(x + 0.5), yi = (int)floor(y + 0.5);
    sm.rx[i] = (int16_t)xi;
    sm.ry[i] = (int16_t)yi;
    sm.score[i] = pre[(size_t)f * RCC_MAX_KEPT + i].score;
    bool v = (xi >= 5 && yi >= 5 && xi < w - 5 && yi < h - 5);
    if (v && xj_check) v = ring_ok(b, w, h, xi, yi);
    sm.ok[i] = v ? 1 : 0;
  }
  __syncthreads();
  // (branch-free and unrolled: with an early exit every one of the ~80 iterations waited for its own LDS round trip:
  // 31 -> 18 us of this one-wave-per-frame kernel; the decision is the same)
  for (int i = lane; i < n; i += 64) {
    bool keep = sm.ok[i];
    const int xi = sm.rx[i], yi = sm.ry[i], si = sm.score[i];
#pragma unroll 8
    for (int j = 0; j < n; ++j) {
      const int dx = abs((int)sm.rx[j] - xi), dy = abs((int)sm.ry[j] - yi), sj = sm.score[j];
      const bool beats = sm.ok[j] && (j != i) && (dx <= dedupe_radius) && (dy <= dedupe_radius) && (sj > si || (sj == si && j < i));
      keep = keep && !beats;
    }
    sm.keep[i] = keep ? 1 : 0;
  }
  __syncthreads();
  // ordered compaction
  int m = 0;
  for (int base = 0; base < n; base += 64) {
    int i = base + lane;
    bool k = (i < n) && sm.keep[i];
    unsigned long long bal = __ballot(k);
    if (k) {
      int o = m + __popcll(bal & ((1ull << lane) - 1ull));
      sm.px[o] = sm.rx[i];
      sm.py[o] = sm.ry[i];
      const double x = pre_xy[((size_t)f * RCC_MAX_KEPT + i) * 2], y = pre_xy[((size_t)f * RCC_MAX_KEPT + i) * 2 + 1];
      sm.xy[2 * o] = x;
      sm.xy[2 * o + 1] = y;
      rcc_cand e;
      e.x = sm.rx[i]; e.y = sm.ry[i]; e.score = sm.score[i];
      kept_out[(size_t)f * RCC_MAX_KEPT + o] = e;
      kept_xy_out[((size_t)f * RCC_MAX_KEPT + o) * 2] = x;
      kept_xy_out[((size_t)f * RCC_MAX_KEPT + o) * 2 + 1] = y;
    }
    m += __popcll(bal);
  }
  __syncthreads();
  const int nk = m;
  if (lane == 0) out->nkept = nk;

  // ---- a6 board indexing
  const int need = cols * rows;
  const bool small = ((long long)w * w + (long long)h * h) < (1ll << 24) && w < 4096 && h < 4096;     // wave-uniform: see nearest_free
  bool found_board = false;
  if (target_kind == RCC_TARGET_CHECKERBOARD && nk >= need && nk <= RCC_MAX_KEPT && cols >= 2 && rows >= 2 &&
      cols <= GBOARD && rows <= GBOARD && need <= RCC_MAX_BOARD_CORNERS) {
    long long sx = 0, sy = 0;
    for (int i = lane; i < nk; i += 64) { sx += sm.px[i]; sy += sm.py[i]; sm.taken[i] = 0; }
    sx = wave_sum_i64(sx);
    sy = wave_sum_i64(sy);
    __syncthreads();
    int seeds[8];
    int nseeds = 0;
    for (int s = 0; s < 8 && s < nk; ++s) {
      unsigned long long best = ~0ull;
      for (int i = lane; i < nk; i += 64) {
        if (sm.taken[i]) continue;
        long long ex = (long long)nk * sm.px[i] - sx, ey = (long long)nk * sm.py[i] - sy;
        unsigned long long key = ((unsigned long long)(ex * ex + ey * ey) << 8) | (unsigned long long)i;
        best = key < best ? key : best;
      }
      best = wave_min_u64(best);
      int bi = (int)(best & 255ull);
      if (lane == 0) sm.taken[bi] = 1;
      __syncthreads();
      seeds[nseeds++] = bi;
    }

    for (int si = 0; si < nseeds && !found_board; ++si) {
      const int s = seeds[si];
      __syncthreads();
      for (int i = lane; i < nk; i += 64) sm.used[i] = (i == s) ? 1 : 0;
      for (int i = lane; i < GW * GW; i += 64) sm.lab[i] = -1;
      __syncthreads();
      const long long sxp = sm.px[s], syp = sm.py[s];
      long long dd;
      const int n1 = small ? nearest_free<true>(sm, nk, lane, sxp, syp, &dd) : nearest_free<false>(sm, nk, lane, sxp, syp, &dd);
      if (n1 < 0) continue;
      const long long ux = sm.px[n1] - sxp, uy = sm.py[n1] - syp;
      const long long uu = ux * ux + uy * uy;
      unsigned long long best = ~0ull;
      for (int k = lane; k < nk; k += 64) {
        if (k == s || k == n1) continue;
        long long wx = sm.px[k] - sxp, wy = sm.py[k] - syp;
        long long cr = ux * wy - uy * wx;
        long long wwv = wx * wx + wy * wy;
        if (4 * cr * cr < uu * wwv) continue;
        unsigned long long key = ((unsigned long long)wwv << 8) | (unsigned long long)k;
        best = key < best ? key : best;
      }
      best = wave_min_u64(best);
      if (best == ~0ull) continue;
      const int n2 = (int)(best & 255ull);
      const long long vx = sm.px[n2] - sxp, vy = sm.py[n2] - syp;

      int qh = 0, qt = 0;
      if (lane == 0) {
        LAB(0, 0) = (int16_t)s;  sm.qi[0] = 0; sm.qj[0] = 0;
        LAB(1, 0) = (int16_t)n1; sm.qi[1] = 1; sm.qj[1] = 0; sm.used[n1] = 1;
        LAB(0, 1) = (int16_t)n2; sm.qi[2] = 0; sm.qj[2] = 1; sm.used[n2] = 1;
      }
      qt = 3;
      int L = 3;
      __syncthreads();
      while (qh < qt) {
        const int i = sm.qi[qh], j = sm.qj[qh];
        ++qh;
        const int ai = LAB(i, j);
        const long long ax = sm.px[ai], ay = sm.py[ai];
#pragma unroll 1
        for (int d = 0; d < 4; ++d) {
          const int di = (d == 0) ? 1 : (d == 1) ? -1 : 0;
          const int dj = (d == 2) ? 1 : (d == 3) ? -1 : 0;
          const int ti = i + di, tj = j + dj;
          if (ti < -GM + 1 || ti > GM - 1 || tj < -GM + 1 || tj > GM - 1) continue;
          if (LAB(ti, tj) >= 0) continue;
          long long predx = 0, predy = 0, step2 = 0;
          bool have = false;
          const int opp = LAB(i - di, j - dj);
          if (opp >= 0) {
            const long long bx = sm.px[opp], by = sm.py[opp];
            predx = 2 * ax - bx; predy = 2 * ay - by;
            step2 = (ax - bx) * (ax - bx) + (ay - by) * (ay - by);
            have = true;
          }
          if (!have) {
            for (int o = -1; o <= 1 && !have; o += 2) {
              const int oi = di ? 0 : o, oj = di ? o : 0;
              const int c0 = LAB(i + oi, j + oj), c1 = LAB(i + oi + di, j + oj + dj);
              if (c0 >= 0 && c1 >= 0) {
                const long long ex = sm.px[c1] - sm.px[c0], ey = sm.py[c1] - sm.py[c0];
                predx = ax + ex; predy = ay + ey;
                step2 = ex * ex + ey * ey;
                have = true;
              }
            }
          }
          if (!have) {
            const long long ex = di ? ux : vx, ey = di ? uy : vy;
            const int sg = di ? di : dj;
            predx = ax + sg * ex; predy = ay + sg * ey;
            step2 = ex * ex + ey * ey;
          }
          long long dist = 0;
          const int k = small ? nearest_free<true>(sm, nk, lane, predx, predy, &dist) : nearest_free<false>(sm, nk, lane, predx, predy, &dist);
          if (k < 0) continue;
          if (8 * dist > step2) continue;
          if (lane == 0) {
            LAB(ti, tj) = (int16_t)k;
            sm.used[k] = 1;
            sm.qi[qt] = (int16_t)ti;
            sm.qj[qt] = (int16_t)tj;
          }
          ++qt;
          ++L;
          __syncthreads();
        }
      }
      if (L != need) continue;
      // un-shear: first k in 0,1,-1,2,-2,3,-3 whose (i + k*j, j) box is cols x rows or rows x cols
      int found = 0, transpose = 0, imin = 0, jmin = 0, shear = 0;
#pragma unroll 1
      for (int t = 0; t < 7 && !found; ++t) {
        const int k = (t == 0) ? 0 : ((t & 1) ? (t + 1) / 2 : -(t / 2));
        int i0 = 1 << 20, i1 = -(1 << 20), j0 = 1 << 20, j1 = -(1 << 20);
        for (int c = lane; c < GW * GW; c += 64) {
          if (sm.lab[c] < 0) continue;
          int ii = c / GW - GM, jj = c % GW - GM;
          int is = ii + k * jj;
          i0 = min(i0, is); i1 = max(i1, is);
          j0 = min(j0, jj); j1 = max(j1, jj);
        }
        i0 = wave_min_i32(i0); i1 = wave_max_i32(i1);
        j0 = wave_min_i32(j0); j1 = wave_max_i32(j1);
        const int bw = i1 - i0 + 1, bh = j1 - j0 + 1;
        if (bw == cols && bh == rows) { found = 1; transpose = 0; }
        else if (bw == rows && bh == cols) { found = 1; transpose = 1; }
        if (found) { imin = i0; jmin = j0; shear = k; }
      }
      if (!found) continue;
      for (int c = lane; c < GW * GW; c += 64) {
        const int idx = sm.lab[c];
        if (idx < 0) continue;
        int ii = c / GW - GM, jj = c % GW - GM;
        int a = ii + shear * jj - imin, bb = jj - jmin;
        int cc = transpose ? bb : a, rr = transpose ? a : bb;
        sm.tmp[rr * cols + cc] = (int16_t)idx;
      }
      __syncthreads();
      const int i00 = sm.tmp[0], ic = sm.tmp[cols - 1], ir = sm.tmp[(rows - 1) * cols];
      const long long crs = (long long)(sm.px[ic] - sm.px[i00]) * (sm.py[ir] - sm.py[i00]) -
                            (long long)(sm.py[ic] - sm.py[i00]) * (sm.px[ir] - sm.px[i00]);
      const bool flipc = crs < 0;
      for (int k = lane; k < need; k += 64) {
        int r = k / cols, c = k - r * cols;
        sm.t2[k] = sm.tmp[r * cols + (flipc ? cols - 1 - c : c)];
      }
      __syncthreads();
      const int a0 = sm.t2[0], a1 = sm.t2[need - 1];
      const bool rot = (sm.py[a1] < sm.py[a0]) || (sm.py[a1] == sm.py[a0] && sm.px[a1] < sm.px[a0]);
      for (int k = lane; k < need; k += 64) sm.order[k] = rot ? sm.t2[need - 1 - k] : sm.t2[k];
      __syncthreads();
      found_board = true;
    }
  }
  if (found_board) {
    for (int k = lane; k < need; k += 64) {
      const int o = sm.order[k];
      out->px[k][0] = sm.px[o];
      out->px[k][1] = sm.py[o];
      out->xy[k][0] = sm.xy[2 * o];
      out->xy[k][1] = sm.xy[2 * o + 1];
    }
    if (lane == 0) out->ncorners = need;
  } else {
    if (lane == 0) { out->status |= RCC_FRAME_NOT_FOUND; out->ncorners = 0; }
  }
  return found_board;
}

