// grid_frame.h -- the per-frame bodies of stages a4.3 and a6 (see k_grid.hip) as device functions.
//   validate_frame: a4.3, spread over the 256 threads of a block (k_validate, k_grid.hip): one candidate per thread.
//   index_frame:    a6, one wavefront per frame: as its own kernel (k_grid_index) or in front of the board pose solve in
//                   one kernel (k_grid_pnp in k_pnp.hip: both are one-wavefront-per-frame dependency chains, and a frame's
//                   pose needs only that frame's lattice).
// They are separate launches because the pose solver's register budget (one wavefront per SIMD) would otherwise be paid by
// the validation's helper wavefronts too: a 256-thread block with that budget fills a whole CU.
#pragma once
#include "rcc_internal.h"
#include "wave_reduce.h"

#ifndef GTRACE
#define GTRACE(slot)
#define GTRACE_VAL(slot, v)
#endif
#define GM 24
#define GW (2 * GM + 1)
#define GBOARD 16
#define GEXTRA 8            // a6: labels beyond cols x rows that the window rule takes (round 4)

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

// The 16 ring samples are fetched in rounds of independent loads -- all grey values, then all tile levels (always in-image
// addresses) -- instead of 16 dependent level -> grey pairs one after the other.
// a4.3 asks for the junction twice: on the threshold map (no flat sample, four transitions) AND on the grey
// ring against its own mid level ((min + max) >> 1; the ring must span min_contrast): the second test is what keeps the
// salt and pepper that a low min_contrast makes of sensor noise from turning an L-shaped outer corner of the board into a junction.
__device__ __forceinline__ bool ring_ok(const BinSrc& b, int w, int h, int x, int y, int min_contrast)
{
  if (x < 5 || y < 5 || x >= w - 5 || y >= h - 5) return false;
  int v[16], g[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) g[k] = b.grey[(size_t)(y + c_ring16[k][1]) * w + (x + c_ring16[k][0])];
  if (!b.thr) {
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = b.bin[(size_t)(y + c_ring16[k][1]) * w + (x + c_ring16[k][0])];
  } else {
    int lv[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int xx = x + c_ring16[k][0], yy = y + c_ring16[k][1];
      const int band = xx / RCC_BAND_W;
      lv[k] = b.thr[((size_t)band * b.th + (yy >> 2)) * RCC_THR_PITCH + ((xx - band * RCC_BAND_W) >> 2)];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = (lv[k] == 255) ? 127 : (g[k] > lv[k] ? 255 : 0);
  }
  int tr = 0, lo = 255, hi = 0;
  bool any127 = false;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    any127 |= (v[k] == 127);
    tr += (v[k] != v[(k + 1) & 15]);
    lo = min(lo, g[k]); hi = max(hi, g[k]);
  }
  const int mid = (lo + hi) >> 1;
  int trg = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) trg += ((g[k] > mid) != (g[(k + 1) & 15] > mid));
  return !any127 && tr == 4 && (hi - lo >= min_contrast) && trg == 4;
}

#define GRID_NOPOS 0x7FFF7FFFu                    // farther than any radius from every valid position (coordinates < 16384)
#define LABP(i, j) sm.labp[((i) + GM) * GW + ((j) + GM)]

// ---- wave-uniform helpers: values that are the same in every lane live in scalar registers ------------------------------
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// lane l (uniform) of register jj (uniform, 0..3) of a 4-register array: point / queue entry lane + 64 * jj
// J registers per lane hold 64 J values: J = 4 covers the list's 256 entries, J = 1 the common case of at most 64 validated
// points (a board frame keeps ~50), where every select below disappears
template <int J>
__device__ __forceinline__ int pick4(const int (&r)[J], int idx)
{
  const int l = idx & 63;
  if (J == 1) return __builtin_amdgcn_readlane(r[0], l);
  const int jj = idx >> 6;
  return jj == 0 ? __builtin_amdgcn_readlane(r[0], l) : jj == 1 ? __builtin_amdgcn_readlane(r[1 % J], l)
       : jj == 2 ? __builtin_amdgcn_readlane(r[2 % J], l) : __builtin_amdgcn_readlane(r[3 % J], l);
}
template <int J>
__device__ __forceinline__ void put4(int (&r)[J], int idx, int val, int lane)
{
  const bool mine = lane == (idx & 63);
  if (J == 1) { r[0] = mine ? val : r[0]; return; }
  const int jj = idx >> 6;
  if (jj == 0) r[0] = mine ? val : r[0];
  else if (jj == 1) r[1 % J] = mine ? val : r[1 % J];
  else if (jj == 2) r[2 % J] = mine ? val : r[2 % J];
  else r[3 % J] = mine ? val : r[3 % J];
}

// The frame's validated points live in REGISTERS during the lattice growth: lane l holds points l, l + 64, l + 128,
// l + 192 (pxr / pyr), their "used" flags as bits of one dword, and the lattice cell each was given.  The growth is a
// single dependency chain per frame (~90 nearest-point searches one after the other); with the points in LDS every
// search was a chain of LDS round trips (flags, coordinates, labels, queue), 73 us of this kernel.
//
// nearest point with its used bit clear to (qx, qy); ties -> smaller index; -1 if none.
// SMALL (w^2 + h^2 < 2^24: every distance between two image points fits 24 bits): the key (distance << 8 | index) is one
// dword and the wave-wide minimum a 32-bit reduction.  A query outside the image may be farther than 2^24 - 2 from every
// point; its distance saturates there, and such a match is refused by the caller either way (8 * distance > step^2,
// step^2 < 2^24).
template <bool SMALL, int J>
__device__ __forceinline__ int nearest_free(const int (&pxr)[J], const int (&pyr)[J], const unsigned usedm, const int nk, const int lane,
                                            const int qx, const int qy, long long* dist)
{
  if (SMALL) {
    unsigned best = ~0u;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      if (64 * j >= nk) break;                       // uniform
      const int k = lane + 64 * j;
      const int dx = pxr[j] - qx, dy = pyr[j] - qy;
      const unsigned d = min((unsigned)__mul24(dx, dx) + (unsigned)__mul24(dy, dy), 0xFFFFFEu);   // |dx|, |dy| < 2^14 here (SMALL: image below 4096 px, predictions within a step of it)
      const unsigned key = (d << 8) | (unsigned)k;
      const bool ok = (k < nk) && !((usedm >> j) & 1u);
      best = (ok && key < best) ? key : best;
    }
    best = wred::all_reduce32(best, [](unsigned a, unsigned b) { return a < b ? a : b; });
    const unsigned b = (unsigned)uni((int)best);
    if (b == ~0u) return -1;
    *dist = (long long)(b >> 8);
    return (int)(b & 255u);
  }
  unsigned long long best = ~0ull;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    if (64 * j >= nk) break;
    const int k = lane + 64 * j;
    const long long dx = (long long)pxr[j] - qx, dy = (long long)pyr[j] - qy;
    const unsigned long long key = ((unsigned long long)(dx * dx + dy * dy) << 8) | (unsigned long long)k;
    const bool ok = (k < nk) && !((usedm >> j) & 1u);
    best = (ok && key < best) ? key : best;
  }
  best = wave_min_u64(best);
  const unsigned lo = (unsigned)uni((int)(unsigned)best), hi = (unsigned)uni((int)(unsigned)(best >> 32));
  const unsigned long long b = (unsigned long long)lo | ((unsigned long long)hi << 32);
  if (b == ~0ull) return -1;
  *dist = (long long)(b >> 8);
  return (int)(b & 255ull);
}

// ---- a4.3: validation of the refined corners of frame f, NT threads (one candidate per thread and pass): ring tests at the
// rounded refined position, de-duplication among the entries that pass, ordered compaction into the frame's kept lists (global) +
// fc[f].nkept.  The suppressed list may hold up to `pre_stride` (2048) entries -- a cluttered scene -- of which at most `max_kept`
// (<= 256, what the lattice stage is built for) may pass the ring tests: more, and the frame is rejected (RCC_FRAME_KEPT_OVERFLOW).
struct valid_smem {
  int32_t score[RCC_MAX_KEPT];     // of the entries that passed the ring tests, in list order
  uint32_t pos[RCC_MAX_KEPT];      // their packed rounded refined pixel x | y << 16
  uint16_t src[RCC_MAX_KEPT];      // their index in the suppressed list
  uint8_t keep[RCC_MAX_KEPT];
  int32_t wcnt[4];
};
template <int NT>
__device__ __forceinline__ void validate_frame(valid_smem& sm, const int f, const int tid,
                                               const uint8_t* __restrict__ bin, const uint8_t* __restrict__ grey,
                                               const uint8_t* __restrict__ thr, int nbands, int w, int h,
                                               const rcc_cand* __restrict__ pre, const int32_t* __restrict__ npre,
                                               const double* __restrict__ pre_xy, int pre_stride, int max_kept, int xj_check, int min_contrast, int dedupe_radius,
                                               rcc_frame_corners* __restrict__ fc,
                                               rcc_cand* __restrict__ kept_out, double* __restrict__ kept_xy_out)
{
  static_assert(NT == 64 || NT == 256, "one or four wavefronts");
  const int lane = tid & 63, wv = tid >> 6;
  rcc_frame_corners* out = fc + f;
  if (out->status != 0) return;         // overflow flagged by the list stage: the frame yields nothing
  BinSrc b;
  b.bin = bin ? bin + (size_t)f * w * h : nullptr;
  b.grey = grey + (size_t)f * w * h;
  b.thr = thr ? thr + (size_t)f * nbands * (h >> 2) * RCC_THR_PITCH : nullptr;
  b.w = w; b.th = h >> 2;
  const int n = npre[f];
  if (max_kept > RCC_MAX_KEPT) max_kept = RCC_MAX_KEPT;

  // ring tests, and the entries that pass gathered in list order (ballot + per-wave counts)
  int m = 0;
  for (int base = 0; base < n; base += NT) {
    const int i = base + tid;
    bool v = false;
    int xi = 0, yi = 0;
    if (i < n) {
      const double x = pre_xy[((size_t)f * pre_stride + i) * 2], y = pre_xy[((size_t)f * pre_stride + i) * 2 + 1];
      xi = (int)floor(x + 0.5); yi = (int)floor(y + 0.5);
      v = (xi >= 5 && yi >= 5 && xi < w - 5 && yi < h - 5);
      if (v && xj_check) v = ring_ok(b, w, h, xi, yi, min_contrast);
    }
    const unsigned long long bal = __ballot(v);
    int before = 0, total = __popcll(bal);
    if (NT > 64) {
      if (lane == 0) sm.wcnt[wv] = total;
      __syncthreads();
      total = 0;
#pragma unroll
      for (int q = 0; q < NT / 64; ++q) { const int c = sm.wcnt[q]; before += (q < wv) ? c : 0; total += c; }
    }
    if (m + total > max_kept) {          // block-uniform: more validated points than the lattice stage takes
      if (tid == 0) { out->status = RCC_FRAME_KEPT_OVERFLOW; out->nkept = 0; }
      return;
    }
    if (v) {
      const int o = m + before + __popcll(bal & ((1ull << lane) - 1ull));
      sm.pos[o] = (unsigned)xi | ((unsigned)yi << 16);
      sm.score[o] = pre[(size_t)f * pre_stride + i].score;
      sm.src[o] = (uint16_t)i;
    }
    m += total;
    if (NT > 64) __syncthreads();        // wcnt is rewritten by the next pass
  }
  __syncthreads();
  // de-duplication among the m entries that passed: entry a goes if another within +-dedupe_radius has a larger score (or the same
  // score and a smaller index -- the gathered order is the list's).  Branch-free and unrolled (an early exit made every iteration
  // wait for its own LDS round trip).
  for (int a = tid; a < m; a += NT) {
    const unsigned pa = sm.pos[a];
    const int xa = (int)(pa & 0xFFFFu), ya = (int)(pa >> 16), sa = sm.score[a];
    bool keep = true;
#pragma unroll 8
    for (int j = 0; j < m; ++j) {
      const unsigned pj = sm.pos[j];
      const int sj = sm.score[j];
      const int dx = abs((int)(pj & 0xFFFFu) - xa), dy = abs((int)(pj >> 16) - ya);
      const bool beats = (j != a) && (dx <= dedupe_radius) && (dy <= dedupe_radius) && (sj > sa || (sj == sa && j < a));
      keep = keep && !beats;
    }
    sm.keep[a] = keep ? 1 : 0;
  }
  __syncthreads();
  // ordered compaction
  int kout = 0;
  for (int base = 0; base < m; base += NT) {
    const int a = base + tid;
    const bool k = (a < m) && sm.keep[a];
    const unsigned long long bal = __ballot(k);
    int before = 0, total = __popcll(bal);
    if (NT > 64) {
      if (lane == 0) sm.wcnt[wv] = total;
      __syncthreads();
      total = 0;
#pragma unroll
      for (int q = 0; q < NT / 64; ++q) { const int c = sm.wcnt[q]; before += (q < wv) ? c : 0; total += c; }
    }
    if (k) {
      const int o = kout + before + __popcll(bal & ((1ull << lane) - 1ull));
      const unsigned pa = sm.pos[a];
      const int i = sm.src[a];
      rcc_cand e;
      e.x = (int16_t)(pa & 0xFFFFu); e.y = (int16_t)(pa >> 16); e.score = sm.score[a];
      kept_out[(size_t)f * RCC_MAX_KEPT + o] = e;
      kept_xy_out[((size_t)f * RCC_MAX_KEPT + o) * 2] = pre_xy[((size_t)f * pre_stride + i) * 2];
      kept_xy_out[((size_t)f * RCC_MAX_KEPT + o) * 2 + 1] = pre_xy[((size_t)f * pre_stride + i) * 2 + 1];
    }
    kout += total;
    if (NT > 64) __syncthreads();        // wcnt is rewritten by the next pass
  }
  if (tid == 0) out->nkept = kout;
}

// ---- a6: board indexing of frame f from its kept lists (validate_frame's output), ONE wavefront.  Returns true when the
// board lattice was found: its corners are then also at sm.xy[2 * sm.order[k]], k = 0..cols*rows-1.
struct grid_smem {
  int32_t px[RCC_MAX_KEPT], py[RCC_MAX_KEPT];     // validated points (rounded refined pixel), in list order
  double xy[2 * RCC_MAX_KEPT];
  int32_t labp[GW * GW];                          // packed coordinates x | y << 16 of the point labelled (i, j); -1: empty
  int16_t tmp[RCC_MAX_KEPT], t2[RCC_MAX_KEPT];
  int32_t order[RCC_MAX_KEPT];
};
// the lattice of a board frame from its validated points (list order in sm.px / sm.py): J = 1 for at most 64 points, else 4
template <int J>
__device__ __forceinline__ bool lattice_board(grid_smem& sm, const int f, const int lane, const int nk, const int cols, const int rows, const int need, const bool small,
                                              const rcc_cand* __restrict__ kept_f /* the frame's validated list (scores of the second seed group) */)
{
  bool found_board = false;
  int pxr[J], pyr[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int k = lane + 64 * j;
    pxr[j] = (k < nk) ? sm.px[k] : 0;
    pyr[j] = (k < nk) ? sm.py[k] : 0;
  }
  long long sx = 0, sy = 0;
#pragma unroll
  for (int j = 0; j < J; ++j) if (lane + 64 * j < nk) { sx += pxr[j]; sy += pyr[j]; }
  sx = wave_sum_i64(sx);
  sy = wave_sum_i64(sy);
  // seeds: the 8 points nearest the centroid, nearest first; seed s is kept in lane s of seedreg (a second group, by score, below)
  unsigned takenm = 0;
  int seedreg = 0;
  const int nseeds = nk < 8 ? nk : 8;
  for (int s = 0; s < nseeds; ++s) {
    unsigned long long best = ~0ull;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int i = lane + 64 * j;
      if (i < nk && !((takenm >> j) & 1u)) {
        const long long ex = (long long)nk * pxr[j] - sx, ey = (long long)nk * pyr[j] - sy;
        const unsigned long long key = ((unsigned long long)(ex * ex + ey * ey) << 8) | (unsigned long long)i;
        best = key < best ? key : best;
      }
    }
    best = wave_min_u64(best);
    const int bi = uni((int)(best & 255ull));
    if (lane == (bi & 63)) takenm |= 1u << (bi >> 6);
    if (lane == s) seedreg = bi;
  }
  GTRACE(1);
  // packed cell offsets of the 3 x 3 neighbourhood a lane < 9 reads: cell (i + lane / 3 - 1, j + lane % 3 - 1)
  const int nb_off = (lane < 9) ? ((lane / 3 - 1) * GW + (lane % 3 - 1)) : 0;

  // round 4: should no centroid seed grow the board (clutter all over the scene pulls the centroid off it), up to 8 more follow: the
  // points with the largest score that have not been tried, largest first, ties to the smaller index -- computed only when needed
  const int nseeds2 = (nk - nseeds) < 8 ? (nk - nseeds) : 8;
  for (int si = 0; si < nseeds + nseeds2 && !found_board; ++si) {
    if (si == nseeds) {                 // wave-uniform
      int scr[J];
#pragma unroll
      for (int j = 0; j < J; ++j) { const int k = lane + 64 * j; scr[j] = (k < nk) ? kept_f[k].score : 0; }
      for (int s2 = 0; s2 < nseeds2; ++s2) {
        unsigned long long best = ~0ull;
#pragma unroll
        for (int j = 0; j < J; ++j) {
          const int i = lane + 64 * j;
          if (i < nk && !((takenm >> j) & 1u)) {
            const unsigned long long key = ((unsigned long long)(~((unsigned)scr[j] ^ 0x80000000u)) << 8) | (unsigned long long)i;     // larger (signed) score = smaller key
            best = key < best ? key : best;
          }
        }
        best = wave_min_u64(best);
        const int bi = uni((int)(best & 255ull));
        if (lane == (bi & 63)) takenm |= 1u << (bi >> 6);
        if (lane == nseeds + s2) seedreg = bi;
      }
    }
    const int s = uni(__builtin_amdgcn_readlane(seedreg, si));
    __syncthreads();
    for (int i = lane; i < GW * GW; i += 64) sm.labp[i] = -1;
    unsigned usedm = (lane == (s & 63)) ? (1u << (s >> 6)) : 0u;
    int cellr[J] = {};                // lattice cell (i + GM) | (j + GM) << 8 of the lane's points, where their used bit is set
    int qreg[J] = {};                  // the growth queue: entry e (a cell, packed as above) in lane e & 63 of register e >> 6
    // give point k the cell (ti, tj): flags, the cell of the point, the label table (packed coordinates of the point)
    auto claim = [&](const int k, const int ti, const int tj, const int packed_xy) {
      const int cellp = (ti + GM) | ((tj + GM) << 8);
      const bool mine = lane == (k & 63);
      const int kj = k >> 6;
      usedm |= mine ? (1u << kj) : 0u;
#pragma unroll
      for (int j = 0; j < J; ++j) cellr[j] = (mine && j == kj) ? cellp : cellr[j];
      if (lane == 0) LABP(ti, tj) = packed_xy;
      return cellp;
    };
    const int sxp = pick4<J>(pxr, s), syp = pick4<J>(pyr, s);
    long long dd;
    const int n1 = small ? nearest_free<true, J>(pxr, pyr, usedm, nk, lane, sxp, syp, &dd) : nearest_free<false, J>(pxr, pyr, usedm, nk, lane, sxp, syp, &dd);
    if (n1 < 0) continue;
    const int n1x = pick4<J>(pxr, n1), n1y = pick4<J>(pyr, n1);
    const long long ux = n1x - sxp, uy = n1y - syp;
    const long long uu = ux * ux + uy * uy;
    unsigned long long best = ~0ull;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int k = lane + 64 * j;
      if (k >= nk || k == s || k == n1) continue;
      const long long wx = pxr[j] - sxp, wy = pyr[j] - syp;
      const long long cr = ux * wy - uy * wx;
      const long long wwv = wx * wx + wy * wy;
      if (4 * cr * cr < uu * wwv) continue;
      const unsigned long long key = ((unsigned long long)wwv << 8) | (unsigned long long)k;
      best = key < best ? key : best;
    }
    best = wave_min_u64(best);
    const unsigned blo = (unsigned)uni((int)(unsigned)best), bhi = (unsigned)uni((int)(unsigned)(best >> 32));
    if (blo == ~0u && bhi == ~0u) continue;
    const int n2 = (int)(blo & 255u);
    const int n2x = pick4<J>(pxr, n2), n2y = pick4<J>(pyr, n2);
    const long long vx = n2x - sxp, vy = n2y - syp;

    put4<J>(qreg, 0, claim(s, 0, 0, sxp | (syp << 16)), lane);
    put4<J>(qreg, 1, claim(n1, 1, 0, n1x | (n1y << 16)), lane);
    put4<J>(qreg, 2, claim(n2, 0, 1, n2x | (n2y << 16)), lane);
    int qh = 0, qt = 3, L = 3;
    __syncthreads();                 // the cleared table and the three labels are in place
    GTRACE(3);
    while (qh < qt) {
      const int cell = pick4<J>(qreg, qh);
      ++qh;
      const int i = (cell & 255) - GM, j = (cell >> 8) - GM;
      // the 3 x 3 neighbourhood of (i, j) in ONE LDS round trip: lane c < 9 holds the packed coordinates of the point at
      // cell (i + c / 3 - 1, j + c % 3 - 1), or -1.  Every rule below reads only these nine cells; a label set for
      // direction d is patched into the register copy, so the later directions see it as the serial definition does.
      int nbv = -1;
      if (lane < 9) nbv = sm.labp[(i + GM) * GW + (j + GM) + nb_off];
      const int a = __builtin_amdgcn_readlane(nbv, 4);
      const int ax = a & 0xFFFF, ay = a >> 16;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int di = (d == 0) ? 1 : (d == 1) ? -1 : 0;
        const int dj = (d == 2) ? 1 : (d == 3) ? -1 : 0;
        const int ti = i + di, tj = j + dj;
        if (ti < -GM + 1 || ti > GM - 1 || tj < -GM + 1 || tj > GM - 1) continue;
        const int tc = (1 + di) * 3 + (1 + dj);
        if (__builtin_amdgcn_readlane(nbv, tc) >= 0) continue;
        int predx = 0, predy = 0;
        long long step2 = 0;
        bool have = false;
        const int opp = __builtin_amdgcn_readlane(nbv, (1 - di) * 3 + (1 - dj));
        if (opp >= 0) {
          const int bx = opp & 0xFFFF, by = opp >> 16;
          predx = 2 * ax - bx; predy = 2 * ay - by;
          step2 = (long long)(ax - bx) * (ax - bx) + (long long)(ay - by) * (ay - by);
          have = true;
        }
        if (!have) {
#pragma unroll
          for (int o = -1; o <= 1; o += 2) {
            const int oi = di ? 0 : o, oj = di ? o : 0;
            const int c0 = __builtin_amdgcn_readlane(nbv, (1 + oi) * 3 + (1 + oj));
            const int c1 = __builtin_amdgcn_readlane(nbv, (1 + oi + di) * 3 + (1 + oj + dj));
            if (!have && c0 >= 0 && c1 >= 0) {
              const int ex = (c1 & 0xFFFF) - (c0 & 0xFFFF), ey = (c1 >> 16) - (c0 >> 16);
              predx = ax + ex; predy = ay + ey;
              step2 = (long long)ex * ex + (long long)ey * ey;
              have = true;
            }
          }
        }
        if (!have) {
          const long long ex = di ? ux : vx, ey = di ? uy : vy;
          const int sg = di ? di : dj;
          predx = ax + sg * (int)ex; predy = ay + sg * (int)ey;
          step2 = ex * ex + ey * ey;
        }
        long long dist = 0;
        const int k = small ? nearest_free<true, J>(pxr, pyr, usedm, nk, lane, predx, predy, &dist) : nearest_free<false, J>(pxr, pyr, usedm, nk, lane, predx, predy, &dist);
        if (k < 0) continue;
        if (8 * dist > step2) continue;
        const int packed = pick4<J>(pxr, k) | (pick4<J>(pyr, k) << 16);
        put4<J>(qreg, qt, claim(k, ti, tj, packed), lane);
        nbv = (lane == tc) ? packed : nbv;
        ++qt;
        ++L;
      }
    }
    GTRACE(2); GTRACE_VAL(7, si * 1000 + L);
    if (L < need || L > need + GEXTRA) continue;
    // un-shear: first k in 0,1,-1,2,-2,3,-3 whose (i + k*j, j) box is cols x rows or rows x cols.  The labelled cells are
    // the cells of the used points (L of them), each in the lane that holds the point.
    int found = 0, transpose = 0, imin = 0, jmin = 0, shear = 0;
    int j0 = 1 << 20, j1 = -(1 << 20);
#pragma unroll
    for (int q = 0; q < J; ++q) if ((usedm >> q) & 1u) { const int jj = (cellr[q] >> 8) - GM; j0 = min(j0, jj); j1 = max(j1, jj); }
    j0 = uni(wave_min_i32(j0)); j1 = uni(wave_max_i32(j1));
    if (L > need) __syncthreads();        // the label table as lane 0 wrote it, for every lane
#pragma unroll 1
    for (int t = 0; t < 7 && !found; ++t) {
      const int k = (t == 0) ? 0 : ((t & 1) ? (t + 1) / 2 : -(t / 2));
      int i0 = 1 << 20, i1 = -(1 << 20);
#pragma unroll
      for (int q = 0; q < J; ++q) if ((usedm >> q) & 1u) {
        const int ii = (cellr[q] & 255) - GM, jj = (cellr[q] >> 8) - GM;
        const int is = ii + k * jj;
        i0 = min(i0, is); i1 = max(i1, is);
      }
      i0 = uni(wave_min_i32(i0)); i1 = uni(wave_max_i32(i1));
      if (L == need) {
        const int bw = i1 - i0 + 1, bh = j1 - j0 + 1;
        if (bw == cols && bh == rows) { found = 1; transpose = 0; }
        else if (bw == rows && bh == cols) { found = 1; transpose = 1; }
        if (found) { imin = i0; jmin = j0; shear = k; }
      } else {
        // round 4: up to GEXTRA labels more than the board has corners (something junction-like next to the board continued a row or
        // a column): the board is the ONE fully labelled cols x rows (or rows x cols) window under the first shear that has any --
        // several: the seed is given up.  Lane c tests cell c of the window; rare path, a handful of windows.
        int wins = 0, wi = 0, wj = 0, wt = 0;
        const int ntr = (cols == rows) ? 1 : 2;
        for (int tr = 0; tr < ntr; ++tr) {
          const int cw = tr ? rows : cols, ch = tr ? cols : rows;
          for (int a0 = i0; a0 + cw - 1 <= i1; ++a0)
            for (int b0 = j0; b0 + ch - 1 <= j1; ++b0) {
              bool ok = true;
              for (int c = lane; c < need; c += 64) {
                const int b = c / cw, a = c - b * cw;
                const int jj = b0 + b, oi = a0 + a - k * jj;
                const bool in = (oi >= -GM) && (oi <= GM);
                ok = ok && in && (sm.labp[((in ? oi : 0) + GM) * GW + (jj + GM)] >= 0);
              }
              if (__ballot(!ok) == 0ull) { if (!wins) { wi = a0; wj = b0; wt = tr; } ++wins; }
            }
        }
        if (wins == 1) { found = 1; transpose = wt; imin = wi; jmin = wj; shear = k; }
        else if (wins > 1) break;
      }
    }
    if (!found) continue;
#pragma unroll
    for (int q = 0; q < J; ++q) if ((usedm >> q) & 1u) {
      const int ii = (cellr[q] & 255) - GM, jj = (cellr[q] >> 8) - GM;
      const int a2 = ii + shear * jj - imin, bb = jj - jmin;
      if (a2 < 0 || bb < 0 || a2 >= (transpose ? rows : cols) || bb >= (transpose ? cols : rows)) continue;     // a label outside the window (L > need)
      const int cc = transpose ? bb : a2, rr = transpose ? a2 : bb;
      sm.tmp[rr * cols + cc] = (int16_t)(lane + 64 * q);
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
  return found_board;
}

__device__ __forceinline__ bool index_frame(grid_smem& sm, const int f, const int lane, int w, int h,
                                            const rcc_cand* __restrict__ kept, const double* __restrict__ kept_xy,
                                            int target_kind, int cols, int rows, rcc_frame_corners* __restrict__ fc)
{
  rcc_frame_corners* out = fc + f;
  if (out->status != 0) return false;   // overflow flagged by the list stage: the frame yields nothing
  const int nk = uni(out->nkept);
  for (int k = lane; k < nk && k < RCC_MAX_KEPT; k += 64) {
    const rcc_cand e = kept[(size_t)f * RCC_MAX_KEPT + k];
    sm.px[k] = e.x; sm.py[k] = e.y;
    sm.xy[2 * k] = kept_xy[((size_t)f * RCC_MAX_KEPT + k) * 2];
    sm.xy[2 * k + 1] = kept_xy[((size_t)f * RCC_MAX_KEPT + k) * 2 + 1];
  }
  __syncthreads();

  // ---- a6 board indexing
  const int need = cols * rows;
  const bool small = ((long long)w * w + (long long)h * h) < (1ll << 24) && w < 4096 && h < 4096;     // wave-uniform: see nearest_free
  bool found_board = false;
  if (target_kind == RCC_TARGET_CHECKERBOARD && nk >= need && nk <= RCC_MAX_KEPT && cols >= 2 && rows >= 2 &&
      cols <= GBOARD && rows <= GBOARD && need <= RCC_MAX_BOARD_CORNERS) {
    const rcc_cand* kf = kept + (size_t)f * RCC_MAX_KEPT;
    found_board = (nk <= 64) ? lattice_board<1>(sm, f, lane, nk, cols, rows, need, small, kf) : lattice_board<4>(sm, f, lane, nk, cols, rows, need, small, kf);
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
