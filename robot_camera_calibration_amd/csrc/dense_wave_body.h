// dense_wave_body.h -- the wave-per-window form of the threshold + corner pass as a device function (the design: k_dense_wave.hip),
// shared by k_dense_wave.hip (the pass as its own kernel) and k_mix.hip (side by side with the ingest pass of the next chunk
// of frames in one launch).
#pragma once
#include "dense_band_body.h"

#ifndef WAVE_DEPTH
#define WAVE_DEPTH 1
#endif
#define WAVE_RING (WAVE_DEPTH + 3)
#ifndef WAVE_DMA_NT
#define WAVE_DMA_NT false
#endif
#ifndef WAVE_PRIO
#define WAVE_PRIO 0           // s_setprio around the corner stages: independent waves have nobody to get ahead of (measured the same with 1)
#endif

// vblock: the job's position in the deal of jobs to XCDs (workgroup id for the stand-alone kernel: its low three bits are the
// XCD the hardware gave the workgroup to, the rest counts that XCD's jobs; k_mix.hip hands in the same thing for a wavefront)
// GANG = 1: every window is a workgroup of its own (the form described above, what the product launches).  GANG = 8
// (rcc_set_dense_gang, measurement only): the eight windows of a band segment are the eight wavefronts of ONE workgroup --
// still no shared rows, no per-row barrier, each wave with its own ring -- but every `syncmask + 1` tile rows the gang
// meets at a barrier, so that no wave runs far ahead of its neighbours: the 64-byte sectors two neighbouring windows share
// (a 256-byte fetch every 244 bytes touches five) are then still cached when the second window asks for them.  Measured on
// 1024 x 1080p: HBM traffic 1.09 x the algorithmic bytes instead of 1.48 x, and 0.83-0.87 ms instead of 0.70 ms (gangs of 2
// and 4 windows: 0.91 / 0.83 ms) -- the pass is bound by vector-instruction issue, not by its fetches, and the free-running
// windows keep the SIMDs evenly loaded; so the re-reads stay (profiles/r03_e_pmc_dense_gang.json).
template <int PRIO, int GANG>
__device__ __forceinline__ void dense_wave_body(const uint8_t* __restrict__ grey, int w, int h, int nbands, int nwin, int nseg, int seg_tiles, int nframes,
                                                int min_contrast, int hthresh, int margin, int cap, int allow_skip, uint8_t* __restrict__ thr_map,
                                                rcc_cand* __restrict__ cand, int32_t* __restrict__ cand_count, int fchunk, int syncmask,
                                                const unsigned vblock /* the job's position in the deal: see below */, const int wib /* GANG > 1: wavefront of the workgroup */,
                                                uint8_t* const ring /* this wavefront's WAVE_RING KiB of LDS, 1 KiB aligned */)
{
  // workgroup -> job, as the band kernel deals them (an XCD takes chunks of consecutive frames along a diagonal); the
  // windows of a band segment are consecutive jobs of one XCD, so the halo columns they share meet in its L2
  const int wpj = (GANG > 1) ? 1 : nwin;                       // jobs (workgroups) per band segment
  const int k = (int)(vblock >> 3);
  int f, wv, band, seg;
  if (fchunk > 0) {
    const int jpf = nbands * nseg * wpj, cj = fchunk * jpf;
    const int grp = k / cj, r = k - grp * cj;
    f = (8 * grp + (((int)(vblock & 7u) - grp) & 7)) * fchunk + r / jpf;
    const int jr = r % jpf;
    wv = (GANG > 1) ? wib : jr % wpj;
    band = (jr / wpj) % nbands; seg = jr / (wpj * nbands);
  } else {
    // fchunk <= 0: the SEGMENT-interleaved deal.  The unit is a band segment of a frame (its windows stay together: they share halo
    // columns in the XCD's L2); unit U = frame * nseg + segment, and of every eight consecutive units each XCD takes one, the
    // assignment rotating with the group -- so an XCD sees every frame (an eighth of it) and every segment position equally often.
    // (With whole frames per XCD the XCDs' loads are sums over 128 frames each, and the launch ends with the busiest XCD.)
    const int wpu = nbands * wpj;
    const int u = k / wpu, jr = k - u * wpu;
    const long long U = 8LL * u + (long long)((((int)(vblock & 7u)) + u) & 7);
    f = (int)(U / nseg); seg = (int)(U - (long long)f * nseg);
    wv = (GANG > 1) ? wib : jr % wpj;
    band = jr / wpj;
  }
  if (f >= nframes) return;
  if (GANG > 1 && wv >= nwin) return;
  const int lane = threadIdx.x & 63;
  const int th = h >> 2;
  const int t0 = seg * seg_tiles;
  const int t1 = min(t0 + seg_tiles, th);
  const int X0 = band * BAND_W, X1 = min(X0 + BAND_W, w);
  if (X0 + wv * STRIP_USE >= X1 || t0 >= t1) return;          // window without band pixels / empty segment (uniform)
  const int xw = X0 + wv * STRIP_USE - 8;                       // first pixel of the window
  const int x0 = xw + 4 * lane;                                 // first pixel of this lane
  const int xl = min(max(x0, 0), w - 4);                        // clamped column (as the strip and band kernels)
  const bool lane_out = (lane >= 2) && (lane <= 62) && (x0 >= X0) && (x0 < X1);
  const int xs = min(max(xw, 0), w - 256);                      // first staged column: 256 bytes that hold every clamped lane
  if (margin < 6) margin = 6;

#ifdef RCC_EXPERIMENTS
  // experiment (scratch/t_overlap.py): read the grey rows of frame f mod g_dense_fmod -- a working set that stays in the Infinity Cache
  // (GANG == 1 only: the otherwise unused syncmask carries the modulus)
  const uint8_t* gf = grey + (size_t)((GANG == 1 && syncmask > 0) ? f % syncmask : f) * w * h;
#else
  const uint8_t* gf = grey + (size_t)f * w * h;
#endif
  const uint64_t ga = (uint64_t)(uintptr_t)gf;
  i32x4 rs_g;
  rs_g.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)ga);
  rs_g.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(ga >> 32) & 0xFFFFu));
  rs_g.z = w * h;
  rs_g.w = 0x00020000;
  uint8_t* bo = thr_map + ((size_t)f * nbands + band) * (size_t)th * RCC_THR_PITCH;
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(bo, 0, th * RCC_THR_PITCH, 0x00020000);
  const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ring);
  const int dk = lane >> 4, di = lane & 15;                     // quarter-wave k moves row k, 16 B per lane
  const int v_edge = xs + 16 * di, v_in = (int)__umul24((unsigned)dk, (unsigned)w) + v_edge;
  const unsigned rd_off = (unsigned)(xl - xs);                  // + 256 k + slot
  const int st_voff = lane_out ? ((x0 - X0) >> 2) : BAND_INVALID;

  // tile row tt -> ring slot; rows clamped to the image (the height is a multiple of 4: a tile row is inside or outside)
  auto issue_dma = [&](int tt, int slot) {
    const bool inside = (tt >= 0) && (tt < th);
    const int soff = __builtin_amdgcn_readfirstlane(inside ? 4 * tt * w : (tt < 0 ? 0 : (h - 1) * w));
    dma_1k<WAVE_DMA_NT>(rs_g, ring_lds + (unsigned)(slot * 1024), inside ? v_in : v_edge, soff);
  };
  auto read_tile = [&](int slot) -> Tile4 {
    const uint8_t* p = ring + slot * 1024 + rd_off;
    Tile4 T;
    T.g0 = *reinterpret_cast<const unsigned*>(p);
    T.g1 = *reinterpret_cast<const unsigned*>(p + 256);
    T.g2 = *reinterpret_cast<const unsigned*>(p + 512);
    T.g3 = *reinterpret_cast<const unsigned*>(p + 768);
    return T;
  };

  SobelRow S0 = { 0, 0, 0, 0 }, S1 = S0, S2 = S0;
  TStat H0 = { 255, 0 }, H1 = H0, H2 = H0;
  RowPipe P;
  P.reset();
  P.w = w; P.h = h; P.t0 = t0; P.t1 = t1; P.margin = margin; P.hthresh = hthresh; P.cap = cap; P.f = f;
  P.cand = cand; P.cand_count = cand_count;
  P.set_lane(x0, lane, lane_out);
  typedef unsigned long long mask64;
  int thrB = 0;
  mask64 flatB = ~0ull;
  const mask64 core_lanes = 0x7FFFFFFFFFFFFFFCull;
  int sf = 0;                                                   // ring slot of tile row t

  auto do_tile = [&](const int t, const TStat& ha, const TStat& hb, TStat& hn,
                     const mask64 Fa, const mask64 Fb, mask64& Fn, SobelRow& sa, SobelRow& sb, SobelRow& sc) {
    if (GANG > 1 && ((t - t0) & syncmask) == 0) __builtin_amdgcn_s_barrier();     // the gang's meeting point (data-wise nothing depends on it)
    // tile row t has landed: the DMA that fetched it has 2 * WAVE_DEPTH - 1 younger operations of this wave
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * WAVE_DEPTH - 1) : "memory");
    const int sd = (sf + WAVE_DEPTH >= WAVE_RING) ? sf + WAVE_DEPTH - WAVE_RING : sf + WAVE_DEPTH;
    const int sb2 = (sf >= 2) ? sf - 2 : sf + WAVE_RING - 2;
    issue_dma(t + WAVE_DEPTH, sd);        // the slot held tile row t-3, last read (and consumed) in iteration t-1
    // ---- FRONT
    hn = tile_stats(read_tile(sf));
    const int dmin = min(ha.hmin, min(hb.hmin, hn.hmin)), dmax = max(ha.hmax, max(hb.hmax, hn.hmax));
    const int range = dmax - dmin;
    const int thrN = dmin + (range >> 1);
    const mask64 flatN = __builtin_amdgcn_ballot_w64(range < min_contrast);
    Fn = allow_skip ? (((t - 1) < t0 - 1) ? ~0ull : flatN) : 0ull;
    // ---- BACK
    const int tau = t - 2;
    const int level = __builtin_amdgcn_inverse_ballot_w64(flatB) ? 255 : thrB;     // tile row tau's level
    if (tau >= t0 - 2) {
      if ((core_lanes & ~(Fa & Fb & Fn)) != 0ull) {
        if (PRIO) __builtin_amdgcn_s_setprio(2);
        const Tile4 B = read_tile(sb2);
        P.row(4 * tau + 0, 0, B.g0, sa, sb, sc, false);
        P.row(4 * tau + 1, 1, B.g1, sb, sc, sa, __builtin_amdgcn_inverse_ballot_w64(Fa));
        P.row(4 * tau + 2, 2, B.g2, sc, sa, sb, false);
        P.row(4 * tau + 3, 3, B.g3, sa, sb, sc, __builtin_amdgcn_inverse_ballot_w64(Fb));
        if (PRIO) __builtin_amdgcn_s_setprio(0);
      } else {
        P.skip();
        dontcare(sa); dontcare(sb); dontcare(sc);
      }
    }
    // the one store of the iteration (dropped where there is nothing to write: the counted wait relies on its presence)
    const bool ok = (tau >= t0) && (tau < t1);
    __builtin_amdgcn_raw_buffer_store_b8((uint8_t)level, rs_b, ok ? st_voff : BAND_INVALID,
                                         __builtin_amdgcn_readfirstlane(ok ? tau * RCC_THR_PITCH : 0), 0);
    thrB = thrN; flatB = flatN;
    sf = (sf + 1 == WAVE_RING) ? 0 : sf + 1;
  };

  int t = t0 - 2;
#pragma unroll
  for (int d = 0; d < WAVE_DEPTH; ++d) issue_dma(t + d, d);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the loop's counted wait only reasons about its own operations
  mask64 F0 = allow_skip ? ~0ull : 0ull, F1 = F0, F2 = F0;
  const int tend = t1 + 2;
  for (;;) {
    do_tile(t, H0, H1, H2, F0, F1, F2, S0, S1, S2);
    if (++t > tend) break;
    do_tile(t, H1, H2, H0, F1, F2, F0, S1, S2, S0);
    if (++t > tend) break;
    do_tile(t, H2, H0, H1, F2, F0, F1, S2, S0, S1);
    if (++t > tend) break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // a DMA still in flight must not land in the next workgroup's LDS
}

