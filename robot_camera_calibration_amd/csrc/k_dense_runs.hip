// k_dense_runs.hip -- the corner half of the two-kernel threshold + corner pass (a4.1): integer Harris response on the
// even lattice, 3x3 lattice maxima, wave-ballot compaction -- only on the (window, tile row) units that can hold a
// candidate.
//
// Why two kernels.  In the fused band kernel (k_dense_band.hip) a window's corner stages run on the wave that owns the
// window, once per tile row, between two workgroup barriers: ~330 vector instructions on the ~35 % of (window, tile row)
// units that are not flat, none on the rest.  The active windows of a frame are neighbours (the target), so one or two
// waves per SIMD carry the whole chain while the others wait at the barrier: a wave alone issues one vector instruction
// per 4 cycles, a gfx950 SIMD retires one per 2 (MI355X_MICROARCH.md, execution model), and the counters show the
// vector pipe ~35 % used with the kernel at 1.16 ms for 0.78 ms of data movement (profiles/r01_o_sq_dense.txt).
// Here the streaming half (k_dense_band<.., SPLIT>) only thresholds and leaves, per (window, tile row), the ballot of
// its lanes' flat flags; this kernel then gives every such window one wave that walks ONLY the active tile rows, with no
// barrier, no LDS and nothing else on its mind -- every resident wave is issuing, five per SIMD.
//
// Same arithmetic, same windows, same masks as the fused kernels (dense_rows.h RowPipe): bit-identical candidates.
//   * job = (frame, segment, band, window); lane l holds pixels x0 + 4 l .. + 3 of the window, lanes 0, 1, 63 halo;
//   * lane L first holds the mask of tile row t0 - 3 + L (rows below t0 - 1: "don't care" = flat, as the fused kernel's
//     warm-up rule); back(tau) is due iff a core lane is not flat in tau - 1, tau or tau + 1 -- one ballot gives the
//     wave its list of rows, and a wave with none exits;
//   * the rows of the next due tile row are in flight while the current one is computed (4 dwords per lane);
//   * a gap in the list leaves the row state "don't care", exactly as a skipped iteration of the fused kernel does.
#include "dense_rows.h"

#define RUNS_MAX_SEG 59            // tile rows per segment: t0 - 3 .. t1 + 1 must fit the 64 lanes

__global__ __launch_bounds__(64) void k_dense_runs(const uint8_t* __restrict__ grey, int w, int h, int nbands, int nseg, int seg_tiles,
                                                    int nframes, int hthresh, int margin, int cap, int allow_skip,
                                                    const unsigned long long* __restrict__ flat, int flat_tp,
                                                    rcc_cand* __restrict__ cand, int32_t* __restrict__ cand_count)
{
  const int lane = threadIdx.x;
  const int job = __builtin_amdgcn_readfirstlane((int)blockIdx.x);
  const int wv = job & 7;
  const int band = (job >> 3) % nbands;
  const int seg = ((job >> 3) / nbands) % nseg;
  const int f = (job >> 3) / (nbands * nseg);
  const int th = h >> 2;
  const int t0 = seg * seg_tiles;
  const int t1 = min(t0 + seg_tiles, th);
  const int X0 = band * RCC_BAND_W, X1 = min(X0 + RCC_BAND_W, w);
  if (X0 + wv * STRIP_USE >= X1 || t0 >= t1) return;           // no band pixels in this window
  const int x0 = X0 + wv * STRIP_USE - 8 + 4 * lane;
  const int xl = min(max(x0, 0), w - 4);
  const bool lane_out = (lane >= 2) && (lane <= 62) && (x0 >= X0) && (x0 < X1);
  if (margin < 6) margin = 6;

  // ---- which tile rows are due
  const unsigned long long CORE = 0x7FFFFFFFFFFFFFFCull;     // lanes 2 .. 62 vote
  const int xrow = t0 - 3 + lane;
  const int nl = t1 - t0 + 4;                                 // lanes 0 .. nl hold rows t0 - 3 .. t1 + 1
  unsigned long long F = ~0ull;
  if (lane >= 2 && lane <= nl) F = allow_skip ? flat[rcc_flat_index(f, band, wv, nbands, flat_tp) + xrow + 1] : 0ull;
  if (!allow_skip && lane < 2) F = 0ull;
  const unsigned flo = (unsigned)F, fhi = (unsigned)(F >> 32);
  const unsigned long long Fm = ((unsigned long long)(unsigned)__shfl_up((int)fhi, 1) << 32) | (unsigned)__shfl_up((int)flo, 1);
  const unsigned long long Fp = ((unsigned long long)(unsigned)__shfl_down((int)fhi, 1) << 32) | (unsigned)__shfl_down((int)flo, 1);
  const bool due = (lane >= 1) && (lane < nl) && (((~(Fm & F & Fp)) & CORE) != 0ull);   // tau = t0 - 2 .. t1
  unsigned long long AR = __ballot(due);
  if (AR == 0ull) return;

  const uint8_t* gf = grey + (size_t)f * w * h;
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(gf), 0, w * h, 0x00020000);
  auto load_tile = [&](int tau) -> Tile4 {
    Tile4 T;
    const int r = 4 * tau;
    T.g0 = __builtin_amdgcn_raw_buffer_load_b32(rs_g, xl, min(max(r, 0), h - 1) * w, 0);
    T.g1 = __builtin_amdgcn_raw_buffer_load_b32(rs_g, xl, min(max(r + 1, 0), h - 1) * w, 0);
    T.g2 = __builtin_amdgcn_raw_buffer_load_b32(rs_g, xl, min(max(r + 2, 0), h - 1) * w, 0);
    T.g3 = __builtin_amdgcn_raw_buffer_load_b32(rs_g, xl, min(max(r + 3, 0), h - 1) * w, 0);
    return T;
  };

  SobelRow S0 = { 0, 0, 0, 0 }, S1 = S0, S2 = S0;
  RowPipe P;
  P.reset();
  P.w = w; P.h = h; P.t0 = t0; P.t1 = t1; P.margin = margin; P.hthresh = hthresh; P.cap = cap; P.f = f;
  P.cand = cand; P.cand_count = cand_count;
  P.set_lane(x0, lane, lane_out);

  int L = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(AR));      // lane index of the row being computed
  AR &= AR - 1ull;
  Tile4 Bc = load_tile(t0 - 3 + L), Bn = Bc;
  int Ln = 0;
  // one due tile row; returns false after the last.  Roles of the Sobel row sets rotate by renaming (period 3), as in
  // the fused kernels; after a gap any assignment will do.
  auto step = [&](SobelRow& sa, SobelRow& sb, SobelRow& sc) -> bool {
    const int tau = t0 - 3 + L;
    const bool more = AR != 0ull;
    if (more) {
      Ln = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(AR));
      AR &= AR - 1ull;
      Bn = load_tile(t0 - 3 + Ln);
    }
    // masks of tile rows tau - 1 (lane L - 1) and tau (lane L)
    const unsigned alo = (unsigned)__builtin_amdgcn_readlane((int)flo, L - 1), ahi = (unsigned)__builtin_amdgcn_readlane((int)fhi, L - 1);
    const unsigned blo = (unsigned)__builtin_amdgcn_readlane((int)flo, L), bhi = (unsigned)__builtin_amdgcn_readlane((int)fhi, L);
    const bool Fa = ((((lane < 32) ? alo : ahi) >> (lane & 31)) & 1u) != 0;
    const bool Fb = ((((lane < 32) ? blo : bhi) >> (lane & 31)) & 1u) != 0;
    P.row(4 * tau + 0, 0, Bc.g0, sa, sb, sc, false);
    P.row(4 * tau + 1, 1, Bc.g1, sb, sc, sa, Fa);     // produces lattice row 4*tau-2, in tile row tau-1
    P.row(4 * tau + 2, 2, Bc.g2, sc, sa, sb, false);
    P.row(4 * tau + 3, 3, Bc.g3, sa, sb, sc, Fb);     // produces lattice row 4*tau, in tile row tau
    if (!more) return false;
    if (Ln != L + 1) {                                 // rows in between are not visited: state is "don't care"
      P.skip();
      dontcare(sa); dontcare(sb); dontcare(sc);
    }
    L = Ln;
    Bc = Bn;
    return true;
  };
  for (;;) {
    if (!step(S0, S1, S2)) break;
    if (!step(S1, S2, S0)) break;
    if (!step(S2, S0, S1)) break;
  }
}

int rcc_dense_allow_skip(const rcc_handle* h);

// segments of the corner kernel: as many as give every wave at most RUNS_MAX_SEG tile rows, and at least 4 per frame so
// that a frame's active windows spread over several waves
int rcc_dense_runs_segments(int th)
{
  int nseg = (th + RUNS_MAX_SEG - 1) / RUNS_MAX_SEG;
  if (nseg < 6 && th >= 6 * 16) nseg = 6;
  return nseg;
}

hipError_t rcc_launch_dense_runs(rcc_handle* h, const uint8_t* d_grey, int nframes, const unsigned long long* d_flat, int flat_tp,
                                 rcc_cand* d_cand, int32_t* d_cand_count, hipStream_t s)
{
  const rcc_config& c = h->cfg;
  const int w = c.width, ht = c.height, th = ht >> 2;
  const int nbands = (w + RCC_BAND_W - 1) / RCC_BAND_W;
#ifdef RCC_EXPERIMENTS
  static const int nseg_env = getenv("RCC_RUNS_NSEG") ? atoi(getenv("RCC_RUNS_NSEG")) : 0;
#else
  const int nseg_env = 0;
#endif
  int nseg = nseg_env > 0 ? nseg_env : rcc_dense_runs_segments(th);
  int seg_tiles = (th + nseg - 1) / nseg;
  if (seg_tiles > RUNS_MAX_SEG) { seg_tiles = RUNS_MAX_SEG; }
  nseg = (th + seg_tiles - 1) / seg_tiles;
  const long long njobs = (long long)nframes * nseg * nbands * 8;
  hipLaunchKernelGGL(k_dense_runs, dim3((unsigned)njobs), dim3(64), 0, s, d_grey, w, ht, nbands, nseg, seg_tiles, nframes,
                     c.harris_thresh, c.cand_margin, c.max_candidates, rcc_dense_allow_skip(h), d_flat, flat_tp, d_cand, d_cand_count);
  return hipGetLastError();
}
