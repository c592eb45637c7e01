// dense_band_body.h -- the band form of the threshold + corner pass as a device function (see k_dense_band.hip for
// the design), shared by k_dense_band.hip (the pass as its own kernel) and k_mix.hip (side by side with the ingest pass
// of the next chunk in one launch).  `job` = (frame, segment, band) index; `lds` = the workgroup's staging memory.
#pragma once
#include "dense_rows.h"

#define RCC_BAND_ST_AUX 2     // nt: the binary image is written once and read by nothing of this kernel
#define BAND_WAVES 8
#define BAND_W RCC_BAND_W
#ifndef BAND_DEPTH
#define BAND_DEPTH 1          // tile rows of DMA in flight ahead of the front stage.  1 since round 2: with the leaner row
                              // pipeline the pass measures 0.93 -> 0.86 ms (compact form, 1024 x 1080p) against depth 2, and the
                              // stage form's ring (4 slots, 48 KB with its output stage) then fits three workgroups per CU
#endif
#define BAND_RING (BAND_DEPTH + 3)   // + the row being read by the front, and the two behind it the back stage reads
#define BAND_OPITCH 2048
#define BAND_OBUF (4 * BAND_OPITCH)
#define BAND_INVALID 0x7FFFFF00          // buffer offset past every frame: the access is dropped

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// 64 lanes x 16 B, global -> LDS at lds_addr + 16 * lane; no VGPR destination.  M0 is saved and restored around
// the instruction (it belongs to the compiler).
template <bool NT = true>
__device__ __forceinline__ void dma_1k(i32x4 rsrc, unsigned lds_addr, int voff, int soff)
{
  unsigned keep;
  // nt: the grey image is read once by this pass and is far larger than the caches -- as a non-temporal stream it does not
  // push the pass's own output lines out (measured with the nt stores below: stage form 1.03 -> 0.99 ms per 1024 x 1080p).
  // NT = false for the wave-per-window kernel, whose neighbouring windows share 128-byte lines: left to the L2, the second
  // window's fetch hits there instead of going to HBM again.
  if (NT)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen nt lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}

// LDS bytes of one workgroup: the ring + the output stage
template <int MODE, int NCH>
struct BandLds { static constexpr int bytes = BAND_RING * NCH * 1024 + (MODE == 2 ? 2 * RCC_THR_PITCH : 2 * BAND_OBUF); };

// MODE 0: the pass; 1: its data movement only (experiment); 2: the pass with the binary image left as the compact
// threshold map (one byte per 4x4 tile: 255 = flat, else the level) -- what rcc_detect_batch needs, 1/16 of the
// output bytes.  PRIO: raise the priority of computing waves.  NCH: 256-B chunks per staged row (8 when the frame
// is one band of at most 2048 columns, else 9: the ring then fits three workgroups per CU).
// SPLIT: the sweep half of the two-kernel form (k_dense_runs.hip): threshold output only -- no corner stages -- plus,
// per (window, tile row), the 64-bit ballot of the lanes' flat flags (`flat`, layout rcc_flat_index), from which the
// corner kernel takes the rows it has to visit and the per-lane response masks.
template <int MODE, int PRIO, int NCH, bool SPLIT = false>
__device__ __forceinline__ void dense_band_body(const uint8_t* __restrict__ grey, int w, int h,
                                                int nbands, int nseg, int seg_tiles, int nframes,
                                                int min_contrast, int hthresh, int margin, int cap, int allow_skip,
                                                uint8_t* __restrict__ bin, rcc_cand* __restrict__ cand,
                                                int32_t* __restrict__ cand_count, const int job, uint8_t* lds /* BAND_LDS<MODE, NCH> bytes, 1 KiB aligned */,
                                                unsigned long long* __restrict__ flat = nullptr, int flat_tp = 0)
{
  constexpr int BAND_SLOT = NCH * 1024;
  constexpr bool THR = (MODE == 2);
  // output stage first (at an LDS address that is a multiple of 16 KiB: its two halves are then told apart by ONE
  // address bit, and "the other half" is an xor on a per-lane offset), the ring behind it (1 KiB aligned for the DMA)
  constexpr int OB_HALF = THR ? RCC_THR_PITCH : BAND_OBUF;
  uint8_t* const obuf = lds;                                // 2 * OB_HALF
  uint8_t* const ring = lds + 2 * OB_HALF;                  // BAND_RING * BAND_SLOT
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int band = job % nbands;
  const int seg = (job / nbands) % nseg;
  const int f = job / (nbands * nseg);
  const int th = h >> 2;
  const int t0 = seg * seg_tiles;
  const int t1 = min(t0 + seg_tiles, th);
  const int X0 = band * BAND_W, X1 = min(X0 + BAND_W, w);     // output columns of the band
  const int XI0 = X0 >= 128 ? X0 - 128 : 0;                    // first staged column
  const int x0 = X0 + wv * STRIP_USE - 8 + 4 * lane;           // first pixel of this lane
  const int xl = min(max(x0, 0), w - 4);                       // clamped column (as the strip kernel)
  const bool lane_out = (lane >= 2) && (lane <= 62) && (x0 >= X0) && (x0 < X1);
  const bool wave_on = (X0 + wv * STRIP_USE) < X1;             // wave-uniform: does this window hold band pixels?
  const uint8_t* gf = grey + (size_t)f * w * h;
  // output: the frame's binary image, or (THR) this band's slice of the frame's compact threshold map
  const int out_bytes = THR ? th * RCC_THR_PITCH : w * h;
  uint8_t* bo = THR ? bin + ((size_t)f * nbands + band) * (size_t)th * RCC_THR_PITCH : bin + (size_t)f * w * h;
  if (margin < 6) margin = 6;
  // SPLIT: this window's row of flat masks; entry x + 1 describes tile row x (-1 .. th: the image's border rows included)
  unsigned long long* const fm = SPLIT ? flat + rcc_flat_index(f, band, wv, nbands, flat_tp) : nullptr;

  // ---- addressing constants
  const uint64_t ga = (uint64_t)(uintptr_t)gf;
  i32x4 rs_g;
  rs_g.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)ga);
  rs_g.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(ga >> 32) & 0xFFFFu));
  rs_g.z = w * h;
  rs_g.w = 0x00020000;
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(bo, 0, out_bytes, 0x00020000);
  const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ring);
  const int dk = lane >> 4, di = lane & 15;                    // quarter-wave k moves row k, 16 B per lane
  const int dma_col = XI0 + 256 * wv + 16 * di;                // this wave's chunk = wv (and 8 for wave 0)
  const bool need_c8 = (NCH > 8) && (wv == 0) && (min(w - 4, X0 + (BAND_WAVES - 1) * STRIP_USE - 8 + 252) - XI0 >= 2048);
  const int rel = xl - XI0;
  const unsigned rd_off = (unsigned)((rel >> 8) * 1024 + (rel & 255));        // + 256 k + slot
  const unsigned wr_off = (unsigned)(lane_out ? (THR ? (x0 - X0) >> 2 : (x0 - X0)) : 0);   // + BAND_OPITCH k + buffer (THR: tile index)
  const int fl_col = X0 + 256 * wv + 16 * di;                  // flush: this lane's 16 output bytes of row dk
  const bool fl_ok = fl_col < X1;
  const unsigned fl_rd = (unsigned)(dk * BAND_OPITCH + 256 * wv + 16 * di);
  const int fl_voff = fl_ok ? dk * w + fl_col : BAND_INVALID;

  // tile row tt -> ring; rows clamped to the image (as load_row of the strip kernel).  The image height is a multiple of 4
  // (rcc_dense_band_supported), so a tile row lies wholly inside the image or wholly outside: inside, lane (dk, di) reads
  // row 4 tt + dk -- a per-lane constant plus a scalar row offset; outside, every lane reads the clamped row (0 or h - 1).
  const int v_in = (int)__umul24((unsigned)dk, (unsigned)w) + dma_col, v_edge = dma_col;
  auto issue_dma = [&](int tt, int slot) {
    const bool inside = (tt >= 0) && (tt < th);                                       // scalar
    const int soff = __builtin_amdgcn_readfirstlane(inside ? 4 * tt * w : (tt < 0 ? 0 : (h - 1) * w));
    const int voff = inside ? v_in : v_edge;
    dma_1k(rs_g, ring_lds + (unsigned)(slot * BAND_SLOT + wv * 1024), voff, soff);
    if (need_c8) dma_1k(rs_g, ring_lds + (unsigned)(slot * BAND_SLOT + 8 * 1024), voff + 2048, soff);
  };
  // the staged output of tile row tt (written one iteration ago) -> global, whole lines: four image rows of 256
  // bytes per wave, or (THR) the band's 512-byte map row by waves 0 and 1.  Every wave issues the store (dropped
  // through an out-of-range offset where it has nothing to write): the counted wait relies on it.
  auto flush = [&](int tt, unsigned ob_fl) {
    const bool ok = (tt >= t0) && (tt < t1);                   // scalar
    if (THR) {
      const unsigned q = *reinterpret_cast<const unsigned*>(obuf + ob_fl);
      __builtin_amdgcn_raw_buffer_store_b32(q, rs_b, (ok && wv < 2) ? 256 * wv + 4 * lane : BAND_INVALID,
                                            __builtin_amdgcn_readfirstlane(ok ? tt * RCC_THR_PITCH : 0), 0);
    } else {
      const u32x4 q = *reinterpret_cast<const u32x4*>(obuf + ob_fl);
      __builtin_amdgcn_raw_buffer_store_b128(q, rs_b, ok ? fl_voff : BAND_INVALID, __builtin_amdgcn_readfirstlane(ok ? 4 * tt * w : 0), RCC_BAND_ST_AUX);
    }
  };
  auto read_tile = [&](int slot) -> Tile4 {
    const uint8_t* p = ring + slot * BAND_SLOT + rd_off;
    Tile4 T;
    T.g0 = *reinterpret_cast<const unsigned*>(p);
    T.g1 = *reinterpret_cast<const unsigned*>(p + 256);
    T.g2 = *reinterpret_cast<const unsigned*>(p + 512);
    T.g3 = *reinterpret_cast<const unsigned*>(p + 768);
    return T;
  };
  auto stage_thr = [&](unsigned ob_wr, int level) {     // THR: one byte per lane = its tile of this tile row
    if (lane_out) obuf[ob_wr] = (uint8_t)level;
  };
  auto stage_out = [&](unsigned ob_wr, unsigned v0, unsigned v1, unsigned v2, unsigned v3) {
    if (THR) return;
    if (lane_out) {
      uint8_t* p = obuf + ob_wr;
      *reinterpret_cast<unsigned*>(p) = v0;
      *reinterpret_cast<unsigned*>(p + BAND_OPITCH) = v1;
      *reinterpret_cast<unsigned*>(p + 2 * BAND_OPITCH) = v2;
      *reinterpret_cast<unsigned*>(p + 3 * BAND_OPITCH) = v3;
    }
  };

  // ---- pipeline state (roles rotate by renaming, period 3: see the strip kernel)
  SobelRow S0 = { 0, 0, 0, 0 }, S1 = S0, S2 = S0;
  TStat H0 = { 255, 0 }, H1 = H0, H2 = H0;
  RowPipe P;
  P.reset();
  P.w = w; P.h = h; P.t0 = t0; P.t1 = t1; P.margin = margin; P.hthresh = hthresh; P.cap = cap; P.f = f;
  P.cand = cand; P.cand_count = cand_count;
  P.set_lane(x0, lane, lane_out);
#ifdef RCC_BAND_TRACE
  long long tr_wait = 0, tr_act = 0; int tr_n = 0, tr_na = 0; const long long tr_0 = wall_clock64();
#endif
  int thrB = 0;
  // the flat flags are kept as 64-bit ballots in scalar registers: the skip vote is scalar arithmetic on them, and a lane
  // reads its own bit back as a select condition (inverse ballot), never as a 0 / 1 value in a vector register
  typedef unsigned long long mask64;
  mask64 flatB = ~0ull;
  const mask64 core_lanes = 0x7FFFFFFFFFFFFFFCull;          // lanes 2..62 (lane_core)
  int sf = 0;                                                  // ring slot of tile row t (scalar)
  // staging half of iteration t is t & 1, the half being flushed the other one: two per-lane offsets, each flipped by an
  // xor per iteration (offsets within a half stay below OB_HALF, the stage sits on a 2 * OB_HALF boundary)
  unsigned ob_wr = wr_off + (unsigned)(((t0 - 2) & 1) * OB_HALF);
  unsigned ob_fl = (THR ? (unsigned)((256 * wv + 4 * lane) & (RCC_THR_PITCH - 1)) : fl_rd) + (unsigned)(((t0 - 3) & 1) * OB_HALF);

  // One iteration t (FRONT on tile row t, BACK on tile row tau = t-2: see the strip kernel for the skip rule)
  auto do_tile = [&](const int t, const TStat& ha, const TStat& hb, TStat& hn,
                     const mask64 Fa, const mask64 Fb, mask64& Fn, SobelRow& sa, SobelRow& sb, SobelRow& sc) {
    // tile row t has landed in LDS for every wave (5 = the operations each wave has issued since its DMA of
    // tile row t), every wave has finished iteration t-1, and its LDS writes are visible
#ifdef RCC_BAND_TRACE
    const long long tr_a = wall_clock64();
#endif
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * BAND_DEPTH - 1) : "memory");
#ifdef RCC_BAND_TRACE
    const long long tr_b = wall_clock64(); tr_wait += tr_b - tr_a; ++tr_n;
#endif
    const int sd = (sf + BAND_DEPTH >= BAND_RING) ? sf + BAND_DEPTH - BAND_RING : sf + BAND_DEPTH;
    const int sb2 = (sf >= 2) ? sf - 2 : sf + BAND_RING - 2;
    issue_dma(t + BAND_DEPTH, sd);       // the slot held tile row t-3, last read in iteration t-1
    flush(t - 3, ob_fl);
    const unsigned ob = ob_wr;
    ob_fl ^= (unsigned)OB_HALF;
    ob_wr ^= (unsigned)OB_HALF;
    if (MODE == 1) {
      const Tile4 C = read_tile(sf), B = read_tile(sb2);
      stage_out(ob, B.g0 ^ C.g0, B.g1 ^ C.g1, B.g2 ^ C.g2, B.g3 ^ C.g3);
    } else if (wave_on) {
      // ---- FRONT
      hn = tile_stats(read_tile(sf));
      const int dmin = min(ha.hmin, min(hb.hmin, hn.hmin)), dmax = max(ha.hmax, max(hb.hmax, hn.hmax));
      const int range = dmax - dmin;
      const int thrN = dmin + (range >> 1);
      const mask64 flatN = __builtin_amdgcn_ballot_w64(range < min_contrast);
      Fn = allow_skip ? (((t - 1) < t0 - 1) ? ~0ull : flatN) : 0ull;   // warm-up rows: "don't care" (strip kernel)
      // ---- BACK
      const int tau = t - 2;
      if (THR) stage_thr(ob, __builtin_amdgcn_inverse_ballot_w64(flatB) ? 255 : thrB);            // tile row tau's level (thrB <= 254 when not flat)
      if (SPLIT) {
        // true flatness of tile row t-1, for the rows this segment owns (every row is written by exactly one segment)
        const int xr = t - 1;
        if ((xr >= t0 && xr < t1) || (xr == -1 && t0 == 0) || (xr == th && t1 == th)) {
          if (lane == 0) fm[xr + 1] = flatN;
        }
        if (!THR && tau >= t0 - 2) {
          if (wave_any(lane_out && !__builtin_amdgcn_inverse_ballot_w64(flatB))) {
            const Tile4 B = read_tile(sb2);
            const Thr4 thr(thrB, __builtin_amdgcn_inverse_ballot_w64(flatB));
            stage_out(ob, thr(B.g0), thr(B.g1), thr(B.g2), thr(B.g3));
          } else {
            stage_out(ob, 0x7F7F7F7Fu, 0x7F7F7F7Fu, 0x7F7F7F7Fu, 0x7F7F7F7Fu);
          }
        }
      } else if (tau >= t0 - 2) {
        if ((core_lanes & ~(Fa & Fb & Fn)) != 0ull) {   // halo lanes do not vote: nothing they hold reaches an output
          if (PRIO) __builtin_amdgcn_s_setprio(2);          // the wave on the critical path of this iteration
          const Tile4 B = read_tile(sb2);
          if (!THR) {
            const Thr4 thr(thrB, __builtin_amdgcn_inverse_ballot_w64(flatB));
            stage_out(ob, thr(B.g0), thr(B.g1), thr(B.g2), thr(B.g3));
          }
          P.row(4 * tau + 0, 0, B.g0, sa, sb, sc, false);
          P.row(4 * tau + 1, 1, B.g1, sb, sc, sa, __builtin_amdgcn_inverse_ballot_w64(Fa));     // produces lattice row 4*tau-2, in tile row tau-1
          P.row(4 * tau + 2, 2, B.g2, sc, sa, sb, false);
          P.row(4 * tau + 3, 3, B.g3, sa, sb, sc, __builtin_amdgcn_inverse_ballot_w64(Fb));     // produces lattice row 4*tau, in tile row tau
          if (PRIO) __builtin_amdgcn_s_setprio(0);
#ifdef RCC_BAND_TRACE
          tr_act += wall_clock64() - tr_b; ++tr_na;
#endif
        } else {
          stage_out(ob, 0x7F7F7F7Fu, 0x7F7F7F7Fu, 0x7F7F7F7Fu, 0x7F7F7F7Fu);
          P.skip();
          dontcare(sa); dontcare(sb); dontcare(sc);
        }
      }
      thrB = thrN; flatB = flatN;
    }
    sf = (sf + 1 == BAND_RING) ? 0 : sf + 1;
  };

  // prologue: tile rows t0-2 .. t0 are fetched and waited for outright (one memory latency per workgroup), so
  // that the counted wait below only ever has to reason about operations issued by the loop itself: the DMA of
  // tile row t >= t0+1 is issued in iteration t-3 and followed by exactly S(t-3) D(t+1) S(t-2) D(t+2) S(t-1).
  // (Padding the prologue with dropped stores instead is fragile: hipcc merged three identical ones into one.)
  int t = t0 - 2;
#pragma unroll
  for (int d = 0; d < BAND_DEPTH; ++d) issue_dma(t + d, d);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  mask64 F0 = allow_skip ? ~0ull : 0ull, F1 = F0, F2 = F0;
  const int tend = t1 + 2;                                // the back stage lags the front by two tile rows
  for (;;) {
    do_tile(t, H0, H1, H2, F0, F1, F2, S0, S1, S2);
    if (++t > tend) break;
    do_tile(t, H1, H2, H0, F1, F2, F0, S1, S2, S0);
    if (++t > tend) break;
    do_tile(t, H2, H0, H1, F2, F0, F1, S2, S0, S1);
    if (++t > tend) break;
  }
  // DMA still in flight must not land in the next workgroup's LDS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef RCC_BAND_TRACE
  if (lane == 0 && (job % 797) == 5)
    printf("job %d wave %d: total %lld  at-barrier %lld  active-path %lld (x10ns)  iterations %d active %d\n", job, wv, wall_clock64() - tr_0, tr_wait, tr_act, tr_n, tr_na);
#endif
}

