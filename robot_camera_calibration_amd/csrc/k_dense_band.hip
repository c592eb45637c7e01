// k_dense_band.hip -- variant 1 (default) of the threshold + corner pass (a3 + a4.1): one 8-wave workgroup
// marches a full-width band of the frame, grey rows staged through an LDS ring by LDS-DMA, binary rows
// staged through LDS so that every global store is whole 128-byte lines.
//
// Same definitions and bit-exact same outputs as k_dense_lds (k_dense.hip) and k_dense_march
// (k_dense_fast.hip); the per-lane arithmetic IS k_dense_march's (dense_rows.h).  What changes is how bytes
// move, because that is what bounded the strip kernel (profiles/r01_g_*):
//   * strip kernel: 12 dword VMEM instructions per wave per tile row (4 loads, 4 re-reads, 4 stores) kept the
//     texture-address FIFO full ~40 % of the time (SQ_VMEM_TA_ADDR_FIFO_FULL), and its 244-byte output spans
//     wrote partial 128-B lines: 0.58 ms per 2.1 GB against 0.36 ms for whole lines (scratch/membench3.hip);
//   * here a wave issues per tile row ONE 1-KiB LDS-DMA load (buffer_load_dwordx4 ... lds: 4 rows x 256 B of
//     its aligned chunk, no VGPRs) and ONE 1-KiB store (4 rows x 256 B, whole lines), the back stage re-reads
//     its rows from LDS instead of L2, and the grey tile buffers leave the register file.
// Layout:
//   * band = output columns [1920 b, 1920 b + 1920) (15 lines); staged image = columns [XI0, XI0 + 2304),
//     XI0 = band start - 128 (0 for the first band); LDS ring slot = one tile row (4 image rows) stored
//     chunk-major [chunk 0..8][row 0..3][256 B], so that one DMA instruction (64 lanes x 16 B, linear in LDS)
//     fills one chunk: lanes 16k..16k+15 fetch row k;
//   * wave j computes window j (pixels band start + 244 j - 8 .. + 256, lanes 2..62 useful) exactly as a strip
//     of the strip kernel does, reading its dwords from the ring at the window's (unaligned) offset;
//   * threshold dwords go to an LDS row buffer; one iteration later wave j stores columns [256 j, 256 j + 256)
//     of the four rows with one dwordx4 store (quarter-wave per row).
// Synchronisation: one s_barrier per tile row.  Ring of BAND_DEPTH + 3 slots, DMA BAND_DEPTH = 2 tile rows ahead
// (1, 2 and 3 measure the same: the kernel is VALU-bound); the wait before the barrier is a COUNTED
// s_waitcnt vmcnt(2 * BAND_DEPTH - 1): every wave issues at least one DMA and exactly one store per iteration (both
// unconditional -- masked lanes / rows get an out-of-range buffer offset, which the hardware drops), so the DMA of
// tile row t has at least 2 * BAND_DEPTH - 1 younger operations when it is waited for (the first tile rows are
// waited for outright in the prologue).  hipcc is kept out of this bookkeeping: the DMA is inline asm, since for
// the builtin it drains vmcnt(0) before every LDS read.
#include "dense_rows.h"

#define BAND_WAVES 8
#define BAND_W RCC_BAND_W
#ifndef BAND_DEPTH
#define BAND_DEPTH 2          // tile rows of DMA in flight ahead of the front stage
#endif
#define BAND_RING (BAND_DEPTH + 3)   // + the row being read by the front, and the two behind it the back stage reads
#define BAND_OPITCH 2048
#define BAND_OBUF (4 * BAND_OPITCH)
#define BAND_INVALID 0x7FFFFF00          // buffer offset past every frame: the access is dropped

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// 64 lanes x 16 B, global -> LDS at lds_addr + 16 * lane; no VGPR destination.  M0 is saved and restored around
// the instruction (it belongs to the compiler).
__device__ __forceinline__ void dma_1k(i32x4 rsrc, unsigned lds_addr, int voff, int soff)
{
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory");
}

// MODE 0: the pass; 1: its data movement only (experiment); 2: the pass with the binary image left as the compact
// threshold map (one byte per 4x4 tile: 255 = flat, else the level) -- what rcc_detect_batch needs, 1/16 of the
// output bytes.  PRIO: raise the priority of computing waves.  NCH: 256-B chunks per staged row (8 when the frame
// is one band of at most 2048 columns, else 9: the ring then fits three workgroups per CU).
template <int MODE, int PRIO, int NCH>
__device__ __forceinline__ void dense_band_body(const uint8_t* __restrict__ grey, int w, int h,
                                                int nbands, int nseg, int seg_tiles, int nframes,
                                                int min_contrast, int hthresh, int margin, int cap, int allow_skip,
                                                uint8_t* __restrict__ bin, rcc_cand* __restrict__ cand,
                                                int32_t* __restrict__ cand_count)
{
  constexpr int BAND_SLOT = NCH * 1024;
  constexpr bool THR = (MODE == 2);
  __shared__ __attribute__((aligned(1024))) uint8_t ring[BAND_RING * BAND_SLOT];
  __shared__ __attribute__((aligned(16))) uint8_t obuf[THR ? 2 * RCC_THR_PITCH : 2 * BAND_OBUF];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int job = blockIdx.x;
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
  const bool lane_core = (lane >= 2) && (lane <= 62);
  const bool wave_on = (X0 + wv * STRIP_USE) < X1;             // wave-uniform: does this window hold band pixels?
  const uint8_t* gf = grey + (size_t)f * w * h;
  // output: the frame's binary image, or (THR) this band's slice of the frame's compact threshold map
  const int out_bytes = THR ? th * RCC_THR_PITCH : w * h;
  uint8_t* bo = THR ? bin + ((size_t)f * nbands + band) * (size_t)th * RCC_THR_PITCH : bin + (size_t)f * w * h;
  if (margin < 6) margin = 6;

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

  // tile row tt -> ring; rows clamped to the image (as load_row of the strip kernel)
  auto issue_dma = [&](int tt, int slot) {
    const int row = __builtin_amdgcn_readfirstlane(4 * tt);
    const int rr = min(max(row + dk, 0), h - 1);
    dma_1k(rs_g, ring_lds + (unsigned)(slot * BAND_SLOT + wv * 1024), (int)__umul24((unsigned)rr, (unsigned)w) + dma_col, 0);
    if (need_c8) dma_1k(rs_g, ring_lds + (unsigned)(slot * BAND_SLOT + 8 * 1024), (int)__umul24((unsigned)rr, (unsigned)w) + dma_col + 2048, 0);
  };
  // the staged output of tile row tt (written one iteration ago) -> global, whole lines: four image rows of 256
  // bytes per wave, or (THR) the band's 512-byte map row by waves 0 and 1.  Every wave issues the store (dropped
  // through an out-of-range offset where it has nothing to write): the counted wait relies on it.
  auto flush = [&](int tt, int buf) {
    const bool ok = (tt >= t0) && (tt < t1);                   // scalar
    if (THR) {
      const unsigned q = *reinterpret_cast<const unsigned*>(obuf + buf * RCC_THR_PITCH + ((256 * wv + 4 * lane) & (RCC_THR_PITCH - 1)));
      __builtin_amdgcn_raw_buffer_store_b32(q, rs_b, (ok && wv < 2) ? 256 * wv + 4 * lane : BAND_INVALID,
                                            __builtin_amdgcn_readfirstlane(ok ? tt * RCC_THR_PITCH : 0), 0);
    } else {
      const u32x4 q = *reinterpret_cast<const u32x4*>(obuf + buf * BAND_OBUF + fl_rd);
      __builtin_amdgcn_raw_buffer_store_b128(q, rs_b, ok ? fl_voff : BAND_INVALID, __builtin_amdgcn_readfirstlane(ok ? 4 * tt * w : 0), 0);
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
  auto stage_thr = [&](int buf, int level) {     // THR: one byte per lane = its tile of this tile row
    if (lane_out) obuf[buf * RCC_THR_PITCH + wr_off] = (uint8_t)level;
  };
  auto stage_out = [&](int buf, unsigned v0, unsigned v1, unsigned v2, unsigned v3) {
    if (THR) return;
    if (lane_out) {
      uint8_t* p = obuf + buf * BAND_OBUF + wr_off;
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
  P.x0 = x0; P.w = w; P.h = h; P.t0 = t0; P.t1 = t1; P.margin = margin; P.hthresh = hthresh; P.cap = cap; P.f = f; P.lane = lane;
  P.lane_out = lane_out; P.cand = cand; P.cand_count = cand_count;
  int thrB = 0, flatB = 1;
  int sf = 0;                                                  // ring slot of tile row t (scalar)

  // One iteration t (FRONT on tile row t, BACK on tile row tau = t-2: see the strip kernel for the skip rule)
  auto do_tile = [&](const int t, const TStat& ha, const TStat& hb, TStat& hn,
                     const int Fa, const int Fb, int& Fn, SobelRow& sa, SobelRow& sb, SobelRow& sc) {
    // tile row t has landed in LDS for every wave (5 = the operations each wave has issued since its DMA of
    // tile row t), every wave has finished iteration t-1, and its LDS writes are visible
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * BAND_DEPTH - 1) : "memory");
    const int sd = (sf + BAND_DEPTH >= BAND_RING) ? sf + BAND_DEPTH - BAND_RING : sf + BAND_DEPTH;
    const int sb2 = (sf >= 2) ? sf - 2 : sf + BAND_RING - 2;
    issue_dma(t + BAND_DEPTH, sd);       // the slot held tile row t-3, last read in iteration t-1
    flush(t - 3, (t - 1) & 1);
    const int ob = t & 1;
    if (MODE == 1) {
      const Tile4 C = read_tile(sf), B = read_tile(sb2);
      stage_out(ob, B.g0 ^ C.g0, B.g1 ^ C.g1, B.g2 ^ C.g2, B.g3 ^ C.g3);
    } else if (wave_on) {
      // ---- FRONT
      hn = tile_stats(read_tile(sf));
      const int dmin = min(ha.hmin, min(hb.hmin, hn.hmin)), dmax = max(ha.hmax, max(hb.hmax, hn.hmax));
      const int range = dmax - dmin;
      const int thrN = dmin + (range >> 1);
      const int flatN = range < min_contrast;
      Fn = allow_skip ? (((t - 1) < t0 - 1) ? 1 : flatN) : 0;   // warm-up rows: "don't care" (strip kernel)
      // ---- BACK
      const int tau = t - 2;
      if (THR) stage_thr(ob, flatB ? 255 : thrB);            // tile row tau's level (thrB <= 254 when not flat)
      if (tau >= t0 - 2) {
        if (__any(lane_core && !(Fa && Fb && Fn))) {   // halo lanes do not vote: nothing they hold reaches an output
          if (PRIO) __builtin_amdgcn_s_setprio(2);          // the wave on the critical path of this iteration
          const Tile4 B = read_tile(sb2);
          if (!THR) {
            const Thr4 thr(thrB, flatB);
            stage_out(ob, thr(B.g0), thr(B.g1), thr(B.g2), thr(B.g3));
          }
          P.row(4 * tau + 0, 0, B.g0, sa, sb, sc, 0);
          P.row(4 * tau + 1, 1, B.g1, sb, sc, sa, Fa);     // produces lattice row 4*tau-2, in tile row tau-1
          P.row(4 * tau + 2, 2, B.g2, sc, sa, sb, 0);
          P.row(4 * tau + 3, 3, B.g3, sa, sb, sc, Fb);     // produces lattice row 4*tau, in tile row tau
          if (PRIO) __builtin_amdgcn_s_setprio(0);
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
  int F0 = allow_skip, F1 = allow_skip, F2 = allow_skip;
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
}

#define BAND_ARGS const uint8_t* __restrict__ grey, int w, int h, int nbands, int nseg, int seg_tiles, int nframes, int min_contrast, \
                  int hthresh, int margin, int cap, int allow_skip, uint8_t* __restrict__ bin, rcc_cand* __restrict__ cand, int32_t* __restrict__ cand_count
#define BAND_PASS grey, w, h, nbands, nseg, seg_tiles, nframes, min_contrast, hthresh, margin, cap, allow_skip, bin, cand, cand_count
template <int MODE, int PRIO, int NCH>
__global__ __launch_bounds__(512) void k_dense_band(BAND_ARGS) { dense_band_body<MODE, PRIO, NCH>(BAND_PASS); }
// the compact-map form needs 42-47 KB of LDS: three workgroups per CU fit if the kernel stays within 80 VGPRs
// (6 waves per SIMD); measured 1.26 -> 1.18 ms per 1024 x 1080p and 1.44 -> 1.32 ms per 256 x 4K with the cap (6-8
// dwords spill)
template <int MODE, int PRIO, int NCH>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_dense_band_occ6(BAND_ARGS) { dense_band_body<MODE, PRIO, NCH>(BAND_PASS); }

bool rcc_dense_band_supported(const rcc_handle* h, const uint8_t* d_grey, const uint8_t* d_bin)
{
  const rcc_config& c = h->cfg;
  return (c.width % 16 == 0) && (c.height % 4 == 0) && c.width >= 64 && c.height >= 8 &&
         ((long long)c.width * c.height < (1LL << 30)) && c.width < (1 << 16) &&
         ((reinterpret_cast<uintptr_t>(d_grey) & 15) == 0) && ((reinterpret_cast<uintptr_t>(d_bin) & 15) == 0);
}

int rcc_dense_allow_skip(const rcc_handle* h);

template <int MODE, int PRIO, int NCH>
static void launch_band(rcc_handle* h, const uint8_t* d_grey, int nframes, uint8_t* d_out, rcc_cand* d_cand, int32_t* d_cand_count,
                        int nbands, int nseg, int seg_tiles, int allow_skip, hipStream_t s)
{
  const rcc_config& c = h->cfg;
  const long long njobs = (long long)nbands * nseg * nframes;
  if constexpr (MODE == 2)
    hipLaunchKernelGGL((k_dense_band_occ6<MODE, PRIO, NCH>), dim3((unsigned)njobs), dim3(512), 0, s, d_grey, c.width, c.height, nbands, nseg, seg_tiles,
                       nframes, c.thr_min_contrast, c.harris_thresh, c.cand_margin, c.max_candidates, allow_skip, d_out, d_cand, d_cand_count);
  else
    hipLaunchKernelGGL((k_dense_band<MODE, PRIO, NCH>), dim3((unsigned)njobs), dim3(512), 0, s, d_grey, c.width, c.height, nbands, nseg, seg_tiles,
                       nframes, c.thr_min_contrast, c.harris_thresh, c.cand_margin, c.max_candidates, allow_skip, d_out, d_cand, d_cand_count);
}

// h->want_thr (set by rcc_detect_batch): write the compact threshold map h->d_thr instead of the binary image
hipError_t rcc_launch_dense_band(rcc_handle* h, const uint8_t* d_grey, int nframes, uint8_t* d_bin,
                                 rcc_cand* d_cand, int32_t* d_cand_count, hipStream_t s)
{
  const rcc_config& c = h->cfg;
  const int w = c.width, ht = c.height, th = ht >> 2;
  const int nbands = (w + BAND_W - 1) / BAND_W;
  // segments: ~4 workgroups per resident slot over the launch, at least 16 tile rows each so that the
  // 4 warm-up / drain iterations stay a small fraction
  int seg_tiles = th;
  static const long long want_mul = getenv("RCC_DENSE_WANT") ? atoll(getenv("RCC_DENSE_WANT")) : 4;
  const long long want = 256LL * 2 * want_mul;
  while (seg_tiles > 16 && (long long)nbands * nframes * ((th + seg_tiles - 1) / seg_tiles) < want) seg_tiles = (seg_tiles + 1) / 2;
  const int nseg = (th + seg_tiles - 1) / seg_tiles;
  const int allow_skip = rcc_dense_allow_skip(h);
  static const int memonly = getenv("RCC_DENSE_MEMONLY") ? atoi(getenv("RCC_DENSE_MEMONLY")) : 0;
  const bool narrow = (nbands == 1) && (w <= 2048);
  const bool thr = h->want_thr && h->d_thr && !memonly;
  h->bin_from_thr = thr ? 1 : 0;
  if (memonly) launch_band<1, 0, 9>(h, d_grey, nframes, d_bin, d_cand, d_cand_count, nbands, nseg, seg_tiles, allow_skip, s);
  else if (thr && narrow) launch_band<2, 1, 8>(h, d_grey, nframes, h->d_thr, d_cand, d_cand_count, nbands, nseg, seg_tiles, allow_skip, s);
  else if (thr) launch_band<2, 1, 9>(h, d_grey, nframes, h->d_thr, d_cand, d_cand_count, nbands, nseg, seg_tiles, allow_skip, s);
  else if (narrow) launch_band<0, 1, 8>(h, d_grey, nframes, d_bin, d_cand, d_cand_count, nbands, nseg, seg_tiles, allow_skip, s);
  else launch_band<0, 1, 9>(h, d_grey, nframes, d_bin, d_cand, d_cand_count, nbands, nseg, seg_tiles, allow_skip, s);
  return hipGetLastError();
}
