// k_dense_fast.hip -- variant 2 of the threshold + corner pass (a3 + a4.1): row marching, one wavefront
// per strip, registers only.  Serves every geometry with width % 4 == 0 and height % 4 == 0; the band kernel
// (k_dense_band.hip, variant 1) is the default where it applies.
//
// Same definitions and bit-exact same outputs as k_dense_lds (k_dense.hip).  Mapping for gfx950:
//   * one wavefront = one vertical strip: 64 lanes x 4 pixels (one dword of the grey row per lane,
//     a 256 B load per row), lanes 0,1 and 63 are halo => 244 useful pixels;
//   * the wave marches down the rows of its segment; every stencil stage keeps its state in
//     registers (dense_rows.h) -- no LDS;
//   * 16-bit packed arithmetic (v_pk_*) for the Sobel stage and the threshold, v_dot2_i32_i16 for
//     the products + horizontal sums of the structure tensor;
//   * candidates leave through one ballot + one atomic per wave.
// Measured limits of this mapping (profiles/): 12 dword VMEM instructions per tile row
// per wave keep the texture-address FIFO full ~40 % of the time, and the 244-byte output spans write partial
// 128-B lines (0.58 ms per 2.1 GB instead of 0.36 ms) -- the band kernel removes both.
#include "dense_rows.h"

template <int MODE>   // 0: the pass; 1: its loads and stores only (experiment: what the access pattern alone costs)
__global__ __launch_bounds__(256) void k_dense_march(const uint8_t* __restrict__ grey, int w, int h,
                                                     int nstrips, int nseg, int seg_tiles, int nframes,
                                                     int min_contrast, int hthresh, int margin, int cap, int allow_skip,
                                                     uint8_t* __restrict__ bin, rcc_cand* __restrict__ cand,
                                                     int32_t* __restrict__ cand_count)
{
  const int lane = threadIdx.x & 63;
  // the job index is wave-uniform: tell the compiler, so strip/segment/frame/row arithmetic is scalar
  const int job = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int njobs = nstrips * nseg * nframes;
  if (job >= njobs) return;
  const int strip = job % nstrips;
  const int seg = (job / nstrips) % nseg;
  const int f = job / (nstrips * nseg);
  const int th = h >> 2;
  const int t0 = seg * seg_tiles;
  const int t1 = min(t0 + seg_tiles, th);
  const int xs = strip * STRIP_USE - 8;
  const int x0 = xs + 4 * lane;                         // first pixel of this lane
  const int xl = min(max(x0, 0), w - 4);                // clamped load column
  const bool lane_out = (lane >= 2) && (lane <= 62) && (x0 >= 0) && (x0 < w);
  const bool lane_core = (lane >= 2) && (lane <= 62);
  const uint8_t* gf = grey + (size_t)f * w * h;
  uint8_t* bo = bin + (size_t)f * w * h;
  if (margin < 6) margin = 6;

  // ---- pipeline state.  Roles rotate by RENAMING: the tile loop is unrolled by three and each
  // unrolled copy gets the register sets in rotated order, so no state is moved between registers
  // (Sobel partial sets rotate every row: 4 rows per tile row => offset 1 per tile row, period 3;
  // grey tile buffers and the tile statistics have period 3 as well).
  SobelRow S0 = { 0, 0, 0, 0 }, S1 = S0, S2 = S0;
  Tile4 T0 = { 0, 0, 0, 0 }, T1 = T0, T2 = T0;
  TStat H0 = { 255, 0 }, H1 = H0, H2 = H0;
  RowPipe P;
  P.reset();
  P.w = w; P.h = h; P.t0 = t0; P.t1 = t1; P.margin = margin; P.hthresh = hthresh; P.cap = cap; P.f = f;
  P.cand = cand; P.cand_count = cand_count;
  P.set_lane(x0, lane, lane_out);

  // addressing: buffer descriptors of the frame's grey / binary image + per-lane byte offset (VGPR, constant)
  // + row offset (SGPR): no vector arithmetic per access (a per-lane 64-bit multiply-add per access would cost
  // more than the whole threshold stage).  Out-of-range never happens (rows clamped, columns clamped).
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(gf), 0, w * h, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(bo, 0, w * h, 0x00020000);
  const int xlu = xl, xou = max(x0, 0);
  auto load_row = [&](int r) -> unsigned {
    const int rr = min(max(r, 0), h - 1);               // scalar
    return __builtin_amdgcn_raw_buffer_load_b32(rs_g, xlu, rr * w, 0);
  };
  auto store_row = [&](int r, unsigned v) { __builtin_amdgcn_raw_buffer_store_b32(v, rs_b, xou, r * w, 0); };

  // One iteration t runs two coupled pipelines:
  //   FRONT (tile row t): statistics of its 4 rows (C), horizontal + vertical dilation => threshold
  //     level and flatness flag of tile row t-1.  N2 prefetches tile row t+2 (two iterations ahead:
  //     with the back stage skipped an iteration is too short to cover HBM latency otherwise).
  //   BACK (tile row tau = t-2): threshold OUTPUT of tau and the Sobel / structure-tensor / response /
  //     selection stages.  A tile whose dilated contrast is below min_contrast cannot hold a candidate
  //     (its 7x7 supports lie in the flat 12x12 neighbourhood, so R <= ((25*gmax^2)>>4)^2 < hthresh --
  //     checked on the host, else allow_skip = 0), and a candidate's comparison against such a
  //     neighbour is decided by hthresh alone.  So when every lane's tile is flat in tile rows tau-1,
  //     tau, tau+1 the corner stages are skipped for tau (and its threshold output is the constant
  //     127, no pixels needed); responses inside flat tiles are masked to INT32_MIN.  Outputs are
  //     unchanged.  The four rows of tau are re-read (L2 hits: the front read them two iterations
  //     ago), one iteration early (Bn -> Bc) so that their latency is covered too.
  //   Fa / Fb / Fn = flatness of tile rows t-3 / t-2 / t-1 (Fn is set here); thrB / flatB = threshold
  //   level and true flatness of tile row t-2 (carried from the previous iteration).
  int thrB = 0;
  bool flatB = true;
  Tile4 Bc = { 0, 0, 0, 0 }, Bn = Bc;
  auto do_tile = [&](const int t, const Tile4& C, Tile4& N2, const TStat& ha, const TStat& hb, TStat& hn,
                     const bool Fa, const bool Fb, bool& Fn, SobelRow& sa, SobelRow& sb, SobelRow& sc) {
    N2.g0 = load_row(4 * t + 8); N2.g1 = load_row(4 * t + 9); N2.g2 = load_row(4 * t + 10); N2.g3 = load_row(4 * t + 11);
    Bn.g0 = load_row(4 * t - 4); Bn.g1 = load_row(4 * t - 3); Bn.g2 = load_row(4 * t - 2); Bn.g3 = load_row(4 * t - 1);   // rows of tau+1
    if (MODE == 1) {
      const int tau = t - 2;
      if ((tau >= t0) && (tau < t1) && lane_out) {
        store_row(4 * tau + 0, Bc.g0 ^ C.g0);
        store_row(4 * tau + 1, Bc.g1 ^ C.g1);
        store_row(4 * tau + 2, Bc.g2 ^ C.g2);
        store_row(4 * tau + 3, Bc.g3 ^ C.g3);
      }
      Bc = Bn;
      return;
    }
    // ---- FRONT
    hn = tile_stats(C);
    const int dmin = min(ha.hmin, min(hb.hmin, hn.hmin)), dmax = max(ha.hmax, max(hb.hmax, hn.hmax));
    const int range = dmax - dmin;
    const int thrN = dmin + (range >> 1);
    const bool flatN = range < min_contrast;
    // warm-up: tile rows below t0-1 produce no lattice row this job needs (the first needed one is
    // 4*t0-2, in tile row t0-1, whose statistics rest on t0-2..t0, all read), so their flag is "don't
    // care" = 1: back(t0-2) then runs iff tile row t0-1 is not flat, back(t0-1) iff t0-1 or t0 is not.
    Fn = allow_skip ? (((t - 1) < t0 - 1) ? true : flatN) : false;
    // ---- BACK
    const int tau = t - 2;
    if (tau >= t0 - 2) {
      const bool out_row = (tau >= t0) && (tau < t1) && lane_out;
      if (wave_any(lane_core && !(Fa && Fb && Fn))) {   // halo lanes do not vote: nothing they hold reaches an output
        if (out_row) {
          const Thr4 thr(thrB, flatB);
          store_row(4 * tau + 0, thr(Bc.g0));
          store_row(4 * tau + 1, thr(Bc.g1));
          store_row(4 * tau + 2, thr(Bc.g2));
          store_row(4 * tau + 3, thr(Bc.g3));
        }
        P.row(4 * tau + 0, 0, Bc.g0, sa, sb, sc, false);
        P.row(4 * tau + 1, 1, Bc.g1, sb, sc, sa, Fa);     // produces lattice row 4*tau-2, in tile row tau-1
        P.row(4 * tau + 2, 2, Bc.g2, sc, sa, sb, false);
        P.row(4 * tau + 3, 3, Bc.g3, sa, sb, sc, Fb);     // produces lattice row 4*tau, in tile row tau
      } else {
        if (out_row) {
#pragma unroll
          for (int k = 0; k < 4; ++k) store_row(4 * tau + k, 0x7F7F7F7Fu);
        }
        P.skip();
        dontcare(sa); dontcare(sb); dontcare(sc);
      }
    }
    thrB = thrN; flatB = flatN;
    Bc = Bn;
  };

  int t = t0 - 2;
  T0.g0 = load_row(4 * t); T0.g1 = load_row(4 * t + 1); T0.g2 = load_row(4 * t + 2); T0.g3 = load_row(4 * t + 3);
  T1.g0 = load_row(4 * t + 4); T1.g1 = load_row(4 * t + 5); T1.g2 = load_row(4 * t + 6); T1.g3 = load_row(4 * t + 7);
  bool F0 = allow_skip != 0, F1 = F0, F2 = F0;
  const int tend = t1 + 2;                                // the back stage lags the front by two tile rows
  for (;;) {
    do_tile(t, T0, T2, H0, H1, H2, F0, F1, F2, S0, S1, S2);
    if (++t > tend) break;
    do_tile(t, T1, T0, H1, H2, H0, F1, F2, F0, S1, S2, S0);
    if (++t > tend) break;
    do_tile(t, T2, T1, H2, H0, H1, F2, F0, F1, S2, S0, S1);
    if (++t > tend) break;
  }
}

bool rcc_dense_march_supported(const rcc_handle* h)
{
  const rcc_config& c = h->cfg;
  return (c.width % 4 == 0) && (c.height % 4 == 0) && c.width >= 8 && c.height >= 8;
}

// flat-tile skip is exact only if a response inside a flat 12x12 neighbourhood stays below the threshold
int rcc_dense_allow_skip(const rcc_handle* h)
{
  const rcc_config& c = h->cfg;
  const int cdiff = c.thr_min_contrast - 1;
  const long long gmax = cdiff > 0 ? ((4LL * cdiff + 7) >> 3) : 0, amax = (25 * gmax * gmax) >> 4;
  return ((h->dense_skip != 0) && (cdiff >= 0) && (amax * amax < (long long)c.harris_thresh)) ? 1 : 0;
}

hipError_t rcc_launch_dense_march(rcc_handle* h, const uint8_t* d_grey, int nframes, uint8_t* d_bin,
                                  rcc_cand* d_cand, int32_t* d_cand_count, hipStream_t s)
{
  const rcc_config& c = h->cfg;
  const int w = c.width, ht = c.height, th = ht >> 2;
  const int nstrips = (w + STRIP_USE - 1) / STRIP_USE;
  // segments: enough jobs to fill the chip (~16 waves per SIMD over the launch), but at least 8 tile
  // rows per segment so the warm-up tile rows stay a small fraction
  int seg_tiles = th;
  const long long want = 256LL * 4 * 16;
  while (seg_tiles > 8 && (long long)nstrips * nframes * ((th + seg_tiles - 1) / seg_tiles) < want) seg_tiles = (seg_tiles + 1) / 2;
  const int nseg = (th + seg_tiles - 1) / seg_tiles;
  const int allow_skip = rcc_dense_allow_skip(h);
  const long long njobs = (long long)nstrips * nseg * nframes;
  const int blocks = (int)((njobs + 3) / 4);
#ifdef RCC_EXPERIMENTS
  static const int memonly = getenv("RCC_DENSE_MEMONLY") ? atoi(getenv("RCC_DENSE_MEMONLY")) : 0;   // data movement only: WRONG results
  if (memonly)
    hipLaunchKernelGGL(k_dense_march<1>, dim3(blocks), dim3(256), 0, s, d_grey, w, ht, nstrips, nseg, seg_tiles, nframes,
                       c.thr_min_contrast, c.harris_thresh, c.cand_margin, c.max_candidates, allow_skip, d_bin, d_cand, d_cand_count);
  else
#endif
    hipLaunchKernelGGL(k_dense_march<0>, dim3(blocks), dim3(256), 0, s, d_grey, w, ht, nstrips, nseg, seg_tiles, nframes,
                       c.thr_min_contrast, c.harris_thresh, c.cand_margin, c.max_candidates, allow_skip, d_bin, d_cand, d_cand_count);
  return hipGetLastError();
}

// ---- counter calibration: streaming copies with a known byte count, one with one dword per lane per access
// (the strip kernel's width) and one with 16 B per lane (the band kernel's), so that FETCH_SIZE / WRITE_SIZE of
// the dense kernels can be priced (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half of the bytes read).
__global__ __launch_bounds__(256) void k_calib_copy_dword(const unsigned* __restrict__ src, unsigned* __restrict__ dst, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
// four 16-byte loads in flight per lane before the first store: with one (a plain grid-stride loop) the copy is bound by
// bytes in flight, 4.9 TB/s; this form reaches what torch's vectorised elementwise kernel does (6.2-6.3 TB/s for 1:1)
__global__ __launch_bounds__(256) void k_calib_copy_x4(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n)
{
  const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
  if (base + 768 < n) {
    const uint4 a = src[base], b = src[base + 256], c = src[base + 512], d = src[base + 768];
    dst[base] = a; dst[base + 256] = b; dst[base + 512] = c; dst[base + 768] = d;
  } else {
    for (size_t i = base; i < n; i += 256) dst[i] = src[i];
  }
}
hipError_t rcc_launch_copy_x4(const void* src, void* dst, size_t nbytes, hipStream_t s)
{
  hipLaunchKernelGGL(k_calib_copy_x4, dim3((unsigned)((nbytes / 16 + 1023) / 1024)), dim3(256), 0, s, (const uint4*)src, (uint4*)dst, nbytes / 16);
  return hipGetLastError();
}
hipError_t rcc_launch_calib_copy(const void* src, void* dst, size_t nbytes, hipStream_t s)
{
  hipLaunchKernelGGL(k_calib_copy_dword, dim3(256 * 16), dim3(256), 0, s, (const unsigned*)src, (unsigned*)dst, nbytes / 4);
  if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | nbytes) & 15) == 0)
    hipLaunchKernelGGL(k_calib_copy_x4, dim3((unsigned)((nbytes / 16 + 1023) / 1024)), dim3(256), 0, s, (const uint4*)src, (uint4*)dst, nbytes / 16);
  return hipGetLastError();
}
