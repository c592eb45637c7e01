// k_dense_fast.hip -- variant 1 of the threshold + corner pass (a3 + a4.1): row marching.
//
// Same definitions and bit-exact same outputs as k_dense_lds (k_dense.hip); needs width % 4 == 0
// and height % 4 == 0 (every BASELINE.json geometry).  Mapping for gfx950:
//   * one wavefront = one vertical strip: 64 lanes x 4 pixels (one dword of the grey row per lane,
//     a 256 B coalesced load per row), lanes 0,1 and 63 are halo => 244 useful pixels;
//   * the wave marches down the rows of its segment; every stencil stage keeps its state in
//     registers (Sobel partials of 3 rows, structure-tensor row sums of 5 rows folded into pair
//     sums, 3 lattice rows of the response, 3 tile rows of threshold statistics) -- no LDS;
//   * neighbouring lanes exchange edge values with whole-wave DPP shifts (wave_shr / wave_shl);
//   * 16-bit packed arithmetic (v_pk_*) for the Sobel stage and the threshold, v_dot2_i32_i16 for
//     the products + horizontal sums of the structure tensor;
//   * candidates leave through one ballot + one atomic per wave.
// HBM traffic = the algorithmic 2 B/px plus halo re-reads (8/248 per row, mostly L2 hits) plus two
// warm-up tile rows per segment.
#include "rcc_internal.h"

typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

#define DPP_FROM_LEFT 0x138   // wave_shr:1 : lane l reads lane l-1
#define DPP_FROM_RIGHT 0x130  // wave_shl:1 : lane l reads lane l+1

__device__ __forceinline__ int from_left(int v, int edge) { return __builtin_amdgcn_update_dpp(edge, v, DPP_FROM_LEFT, 0xf, 0xf, false); }
__device__ __forceinline__ int from_right(int v, int edge) { return __builtin_amdgcn_update_dpp(edge, v, DPP_FROM_RIGHT, 0xf, 0xf, false); }
// zero for lanes without a source (bound_ctrl:0): lets the compiler fold the move into the consumer
__device__ __forceinline__ int from_left0(int v) { return __builtin_amdgcn_update_dpp(0, v, DPP_FROM_LEFT, 0xf, 0xf, true); }
__device__ __forceinline__ int from_right0(int v) { return __builtin_amdgcn_update_dpp(0, v, DPP_FROM_RIGHT, 0xf, 0xf, true); }
__device__ __forceinline__ i16x2 as_i(unsigned v) { return __builtin_bit_cast(i16x2, v); }
__device__ __forceinline__ u16x2 as_u(unsigned v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ unsigned bits(i16x2 v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ unsigned bits(u16x2 v) { return __builtin_bit_cast(unsigned, v); }

struct HSum { int xx0, xy0, yy0, xx2, xy2, yy2; };   // raw 5-px row sums at pixel 0 and pixel 2 of the lane
struct SobelRow { i16x2 dh01, dh23, sh01, sh23; };     // horizontal Sobel partials of one row (pixel pairs 0-1, 2-3)
struct Tile4 { unsigned g0, g1, g2, g3; };             // grey dwords of the four rows of a tile row
struct TStat { int hmin, hmax; };                      // horizontally dilated tile min / max
struct LRow { int r0, r2, rL, rR; };                   // lattice row of responses: own px 0, px 2, left and right neighbour

// ---- job geometry ----------------------------------------------------------------------------------
#define STRIP_USE 244   // useful pixels per wave strip: lanes 2..62 (the left lattice neighbour of the
                        // first useful pixel needs a 5-pixel halo => two halo lanes on the left, one on the right)

template <int MODE>   // 0: the pass; 1: its loads and stores only (experiment: what the access pattern alone costs)
__global__ __launch_bounds__(256) void k_dense_march(const uint8_t* __restrict__ grey, int w, int h,
                                                     int nstrips, int nseg, int seg_tiles, int nframes,
                                                     int min_contrast, int hthresh, int margin, int cap, int allow_skip_exp,
                                                     uint8_t* __restrict__ bin, rcc_cand* __restrict__ cand,
                                                     int32_t* __restrict__ cand_count)
{
  const int allow_skip = allow_skip_exp & 1, exp_al = allow_skip_exp & 2, exp_ns = allow_skip_exp & 4;   // EXPERIMENT bits
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
  const uint8_t* gf = grey + (size_t)f * w * h;       // uniform; per-lane column offset xl is added at the load
  uint8_t* bo = bin + (size_t)f * w * h;
  if (margin < 6) margin = 6;

  // ---- pipeline state.  Roles rotate by RENAMING: the tile loop is unrolled by three and each
  // unrolled copy gets the register sets in rotated order, so no state is moved between registers
  // (Sobel partial sets rotate every row: 4 rows per tile row => offset 1 per tile row, period 3;
  // grey tile buffers prev/cur/next and the tile statistics have period 3 as well).
  SobelRow S0 = { 0, 0, 0, 0 }, S1 = S0, S2 = S0;
  Tile4 T0 = { 0, 0, 0, 0 }, T1 = T0, T2 = T0;
  TStat H0 = { 255, 0 }, H1 = H0, H2 = H0;
  HSum hprev = { 0, 0, 0, 0, 0, 0 }, qa = hprev, qb = hprev;     // qa: pair closed at k=0, qb: pair closed at k=2
  LRow Ra = { INT32_MIN, INT32_MIN, INT32_MIN, INT32_MIN }, Rb = Ra;   // lattice rows y-4, y-2

  // addressing: buffer descriptors of the frame's grey / binary image + per-lane byte offset (VGPR, constant)
  // + row offset (SGPR): no vector arithmetic per access (a per-lane 64-bit multiply-add per access would cost
  // more than the whole threshold stage).  Out-of-range never happens (rows clamped, columns clamped).
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(gf), 0, w * h, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(bo, 0, w * h, 0x00020000);
  const int xlu = xl, xou = exp_al ? strip * 256 + 4 * lane : max(x0, 0);
  auto load_row = [&](int r) -> unsigned {
    const int rr = min(max(r, 0), h - 1);               // scalar
    return __builtin_amdgcn_raw_buffer_load_b32(rs_g, xlu, rr * w, 0);
  };
  auto store_row = [&](int r, unsigned v) { __builtin_amdgcn_raw_buffer_store_b32(v, rs_b, xou, r * w, 0); };
  // "don't care" values that cost no instruction (the compiler may leave anything in the register)
  auto dontcare = [](int& v) { asm volatile("" : "=v"(v)); };
  auto dontcare_s = [&](SobelRow& q) {
    int a, b, c, d; dontcare(a); dontcare(b); dontcare(c); dontcare(d);
    q.dh01 = __builtin_bit_cast(i16x2, a); q.dh23 = __builtin_bit_cast(i16x2, b); q.sh01 = __builtin_bit_cast(i16x2, c); q.sh23 = __builtin_bit_cast(i16x2, d);
  };
  auto dontcare_h = [&](HSum& q) { dontcare(q.xx0); dontcare(q.xy0); dontcare(q.yy0); dontcare(q.xx2); dontcare(q.xy2); dontcare(q.yy2); };

  // one image row: G = its grey dword; (a, b) = Sobel partials of rows r-2, r-1; n receives row r's.
  // rmask != 0: the lane's tile at the lattice row being produced is flat => its response cannot
  // reach hthresh (host-checked bound), so it is replaced by INT32_MIN
  auto do_row = [&](const int r, const int k, const unsigned G, const SobelRow& a, const SobelRow& b, SobelRow& n,
                    const int rmask) {
    // ---- stage A (row r): horizontal Sobel partials, natural-order pixel pairs
    const unsigned GL = (unsigned)from_left((int)G, (int)G), GR = (unsigned)from_right((int)G, (int)G);
    const u16x2 n0 = as_u(__builtin_amdgcn_perm(0u, G, 0x0C010C00u));    // [p0,p1]
    const u16x2 n1 = as_u(__builtin_amdgcn_perm(0u, G, 0x0C030C02u));    // [p2,p3]
    const u16x2 mm = as_u(__builtin_amdgcn_perm(0u, G, 0x0C020C01u));    // [p1,p2]
    const u16x2 lh = as_u(__builtin_amdgcn_perm(GL, G, 0x0C000C07u));    // [p-1,p0]
    const u16x2 rh = as_u(__builtin_amdgcn_perm(GR, G, 0x0C040C03u));    // [p3,p4]
    const i16x2 two = (i16x2)(2);
    n.dh01 = as_i(bits(mm)) - as_i(bits(lh));                            // I[x+1]-I[x-1] for x = p0,p1
    n.dh23 = as_i(bits(rh)) - as_i(bits(mm));
    n.sh01 = as_i(bits(n0)) * two + as_i(bits(lh)) + as_i(bits(mm));     // I[x-1]+2I[x]+I[x+1]
    n.sh23 = as_i(bits(n1)) * two + as_i(bits(mm)) + as_i(bits(rh));
    // ---- stage B (row rho = r-1): gradients
    const i16x2 gx01 = (b.dh01 * two + a.dh01 + n.dh01) >> 3;
    const i16x2 gx23 = (b.dh23 * two + a.dh23 + n.dh23) >> 3;
    const i16x2 gy01 = (n.sh01 - a.sh01) >> 3;
    const i16x2 gy23 = (n.sh23 - a.sh23) >> 3;
    // ---- stage C (row rho): products + horizontal 5-sums at pixels 0 and 2
    const int d0xx = __builtin_amdgcn_sdot2(gx01, gx01, 0, false);
    const int d0xy = __builtin_amdgcn_sdot2(gx01, gy01, 0, false);
    const int d0yy = __builtin_amdgcn_sdot2(gy01, gy01, 0, false);
    const int d1xx = __builtin_amdgcn_sdot2(gx23, gx23, 0, false);
    const int d1xy = __builtin_amdgcn_sdot2(gx23, gy23, 0, false);
    const int d1yy = __builtin_amdgcn_sdot2(gy23, gy23, 0, false);
    const int ax0 = gx01.x, ay0 = gy01.x, ax2 = gx23.x, ay2 = gy23.x;
    const int q0xx = __mul24(ax0, ax0), q0xy = __mul24(ax0, ay0), q0yy = __mul24(ay0, ay0);
    const int q2xx = __mul24(ax2, ax2), q2xy = __mul24(ax2, ay2), q2yy = __mul24(ay2, ay2);
    HSum hc;
    hc.xx0 = d0xx + q2xx + from_left0(d1xx);
    hc.xy0 = d0xy + q2xy + from_left0(d1xy);
    hc.yy0 = d0yy + q2yy + from_left0(d1yy);
    hc.xx2 = d0xx + d1xx + from_right0(q0xx);
    hc.xy2 = d0xy + d1xy + from_right0(q0xy);
    hc.yy2 = d0yy + d1yy + from_right0(q0yy);
    // ---- stage D/E/F: vertical sums on the lattice, response, selection
    if ((k & 1) == 0) {
      // rho = r-1 is odd: close the pair (rho-1, rho) into qa (k = 0) or qb (k = 2)
      HSum& q = (k == 0) ? qa : qb;
      q.xx0 = hprev.xx0 + hc.xx0; q.xy0 = hprev.xy0 + hc.xy0; q.yy0 = hprev.yy0 + hc.yy0;
      q.xx2 = hprev.xx2 + hc.xx2; q.xy2 = hprev.xy2 + hc.xy2; q.yy2 = hprev.yy2 + hc.yy2;
    } else {
      // rho = r-1 is even: 5-row sums centred on y = rho-2 = the two closed pairs + this row
      const int A0 = (qa.xx0 + qb.xx0 + hc.xx0) >> 4, B0 = (qa.xy0 + qb.xy0 + hc.xy0) >> 4, C0 = (qa.yy0 + qb.yy0 + hc.yy0) >> 4;
      const int A2 = (qa.xx2 + qb.xx2 + hc.xx2) >> 4, B2 = (qa.xy2 + qb.xy2 + hc.xy2) >> 4, C2 = (qa.yy2 + qb.yy2 + hc.yy2) >> 4;
      const unsigned tr0 = (unsigned)(A0 + C0), tr2 = (unsigned)(A2 + C2);
      LRow Rn;
      Rn.r0 = rmask ? INT32_MIN : __mul24(A0, C0) - __mul24(B0, B0) - (int)(__umul24(tr0, tr0) >> 4);
      Rn.r2 = rmask ? INT32_MIN : __mul24(A2, C2) - __mul24(B2, B2) - (int)(__umul24(tr2, tr2) >> 4);
      Rn.rL = from_left(Rn.r2, INT32_MIN);
      Rn.rR = from_right(Rn.r0, INT32_MIN);
      hprev = hc;
      // selection on lattice row yc = rho - 4 = r - 5 (rows Ra = yc-2, Rb = yc, Rn = yc+2)
      const int yc = r - 5;
      const bool rowok = (yc >= 4 * t0) && (yc < 4 * t1) && (yc >= margin) && (yc < h - margin);
      if (rowok && __any((Rb.r0 >= hthresh) || (Rb.r2 >= hthresh))) {
        const int xa = x0, xb = x0 + 2;
        bool is0 = lane_out && Rb.r0 >= hthresh && xa >= margin && xa < w - margin &&
                   Rb.r0 > Ra.rL && Rb.r0 > Ra.r0 && Rb.r0 > Ra.r2 && Rb.r0 > Rb.rL &&
                   Rb.r0 >= Rb.r2 && Rb.r0 >= Rn.rL && Rb.r0 >= Rn.r0 && Rb.r0 >= Rn.r2;
        bool is2 = lane_out && Rb.r2 >= hthresh && xb >= margin && xb < w - margin &&
                   Rb.r2 > Ra.r0 && Rb.r2 > Ra.r2 && Rb.r2 > Ra.rR && Rb.r2 > Rb.r0 &&
                   Rb.r2 >= Rb.rR && Rb.r2 >= Rn.r0 && Rb.r2 >= Rn.r2 && Rb.r2 >= Rn.rR;
        const unsigned long long m0 = __ballot(is0), m2 = __ballot(is2);
        const int n0c = __popcll(m0), n2c = __popcll(m2);
        if (n0c + n2c) {
          int basei = 0;
          if (lane == 0) basei = atomicAdd(&cand_count[f], n0c + n2c);
          basei = __shfl(basei, 0);
          const unsigned long long below = (1ull << lane) - 1ull;
          if (is0) {
            int idx = basei + __popcll(m0 & below);
            if (idx < cap) { rcc_cand e; e.x = (int16_t)xa; e.y = (int16_t)yc; e.score = Rb.r0; cand[(size_t)f * cap + idx] = e; }
          }
          if (is2) {
            int idx = basei + n0c + __popcll(m2 & below);
            if (idx < cap) { rcc_cand e; e.x = (int16_t)xb; e.y = (int16_t)yc; e.score = Rb.r2; cand[(size_t)f * cap + idx] = e; }
          }
        }
      }
      Ra = Rb;
      Rb = Rn;
    }
  };

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
  int thrB = 0, flatB = 1;
  Tile4 Bc = { 0, 0, 0, 0 }, Bn = Bc;
  auto do_tile = [&](const int t, const Tile4& C, Tile4& N2, const TStat& ha, const TStat& hb, TStat& hn,
                     const int Fa, const int Fb, int& Fn, SobelRow& sa, SobelRow& sb, SobelRow& sc) {
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
    u16x2 tmn = (u16x2)(255), tmx = (u16x2)(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned G = (k == 0) ? C.g0 : (k == 1) ? C.g1 : (k == 2) ? C.g2 : C.g3;
      const u16x2 n0 = as_u(__builtin_amdgcn_perm(0u, G, 0x0C010C00u));
      const u16x2 n1 = as_u(__builtin_amdgcn_perm(0u, G, 0x0C030C02u));
      tmn = __builtin_elementwise_min(tmn, __builtin_elementwise_min(n0, n1));
      tmx = __builtin_elementwise_max(tmx, __builtin_elementwise_max(n0, n1));
    }
    const int tmin = min((int)tmn.x, (int)tmn.y), tmax = max((int)tmx.x, (int)tmx.y);
    hn.hmin = min(tmin, min(from_left(tmin, tmin), from_right(tmin, tmin)));
    hn.hmax = max(tmax, max(from_left(tmax, tmax), from_right(tmax, tmax)));
    const int dmin = min(ha.hmin, min(hb.hmin, hn.hmin)), dmax = max(ha.hmax, max(hb.hmax, hn.hmax));
    const int range = dmax - dmin;
    const int thrN = dmin + (range >> 1);
    const int flatN = range < min_contrast;
    // warm-up: tile rows below t0-1 produce no lattice row this job needs (the first needed one is
    // 4*t0-2, in tile row t0-1, whose statistics rest on t0-2..t0, all read), so their flag is "don't
    // care" = 1: back(t0-2) then runs iff tile row t0-1 is not flat, back(t0-1) iff t0-1 or t0 is not.
    Fn = allow_skip ? (((t - 1) < t0 - 1) ? 1 : flatN) : 0;
    // ---- BACK
    const int tau = t - 2;
    if (tau >= t0 - 2) {
      const bool out_row = (tau >= t0) && (tau < t1) && (exp_al ? (strip * 256 + 4 * lane < w) : lane_out) && !exp_ns;
      if (__any(!(Fa && Fb && Fn))) {
        if (out_row) {
          // per byte: v > thr <=> v >= thr+1 (thr <= 254 whenever the tile is not flat).  SWAR unsigned byte
          // compare: d = (x|H) - (y&~H) has its per-byte MSB set iff the low 7 bits of x >= those of y (no
          // borrow crosses bytes); where the MSBs of x and y differ x's decides, else d's.
          const unsigned H = 0x80808080u;
          const unsigned y4 = __builtin_amdgcn_perm(0u, (unsigned)(thrB + 1), 0u);   // byte 0 replicated
          const unsigned ylo = y4 & ~H, ny = ~y4;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const unsigned x = (k == 0) ? Bc.g0 : (k == 1) ? Bc.g1 : (k == 2) ? Bc.g2 : Bc.g3;
            const unsigned d = (x | H) - ylo;
            const unsigned xy = x ^ y4;
            const unsigned ge = (((x & ny) & xy) | (d & ~xy)) & H;      // bitfield select on xy
            unsigned o = ge | (ge - (ge >> 7));                          // 0x80 -> 0xFF per byte, no carries
            if (flatB) o = 0x7F7F7F7Fu;
            store_row(4 * tau + k, o);
          }
        }
        do_row(4 * tau + 0, 0, Bc.g0, sa, sb, sc, 0);
        do_row(4 * tau + 1, 1, Bc.g1, sb, sc, sa, Fa);     // produces lattice row 4*tau-2, in tile row tau-1
        do_row(4 * tau + 2, 2, Bc.g2, sc, sa, sb, 0);
        do_row(4 * tau + 3, 3, Bc.g3, sa, sb, sc, Fb);     // produces lattice row 4*tau, in tile row tau
      } else {
        if (out_row) {
#pragma unroll
          for (int k = 0; k < 4; ++k) store_row(4 * tau + k, 0x7F7F7F7Fu);
        }
        Ra.r0 = Ra.r2 = Ra.rL = Ra.rR = INT32_MIN;        // the two lattice rows not produced lie in flat tiles
        Rb = Ra;
        // the Sobel partials and row sums now describe rows that were not read; everything they can still
        // reach is a masked response (see above), so their contents do not matter: no copies at the join
        dontcare_s(sa); dontcare_s(sb); dontcare_s(sc);
        dontcare_h(hprev); dontcare_h(qa); dontcare_h(qb);
      }
    }
    thrB = thrN; flatB = flatN;
    Bc = Bn;
  };

  int t = t0 - 2;
  T0.g0 = load_row(4 * t); T0.g1 = load_row(4 * t + 1); T0.g2 = load_row(4 * t + 2); T0.g3 = load_row(4 * t + 3);
  T1.g0 = load_row(4 * t + 4); T1.g1 = load_row(4 * t + 5); T1.g2 = load_row(4 * t + 6); T1.g3 = load_row(4 * t + 7);
  int F0 = allow_skip, F1 = allow_skip, F2 = allow_skip;
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

bool rcc_dense_fast_supported(const rcc_handle* h)
{
  const rcc_config& c = h->cfg;
  return (c.width % 4 == 0) && (c.height % 4 == 0) && c.width >= 8 && c.height >= 8;
}

hipError_t rcc_launch_dense_fast(rcc_handle* h, const uint8_t* d_grey, int nframes, uint8_t* d_bin,
                                 rcc_cand* d_cand, int32_t* d_cand_count, hipStream_t s)
{
  const rcc_config& c = h->cfg;
  const int w = c.width, ht = c.height, th = ht >> 2;
  const int nstrips = (w + STRIP_USE - 1) / STRIP_USE;
  // segments: enough jobs to fill the chip (>= ~8 waves per SIMD in flight), but at least 8 tile
  // rows per segment so the 3 warm-up tile rows stay a small fraction
  int seg_tiles = th;
  static const long long want_mul = getenv("RCC_DENSE_WANT") ? atoll(getenv("RCC_DENSE_WANT")) : 8;
  const long long want = 256LL * 4 * want_mul;
  while (seg_tiles > 8 && (long long)nstrips * nframes * ((th + seg_tiles - 1) / seg_tiles) < want) seg_tiles = (seg_tiles + 1) / 2;
  const int nseg = (th + seg_tiles - 1) / seg_tiles;
  // flat-tile skip is exact only if a response inside a flat 12x12 neighbourhood stays below the threshold
  const int cdiff = c.thr_min_contrast - 1;
  const long long gmax = cdiff > 0 ? ((4LL * cdiff + 7) >> 3) : 0, amax = (25 * gmax * gmax) >> 4;
  static const int expbits = getenv("RCC_DENSE_EXP") ? atoi(getenv("RCC_DENSE_EXP")) : 0;
  const int allow_skip = ((h->dense_skip != 0) && (cdiff >= 0) && (amax * amax < (long long)c.harris_thresh) ? 1 : 0) | expbits;
  const long long njobs = (long long)nstrips * nseg * nframes;
  const int blocks = (int)((njobs + 3) / 4);
  static const int memonly = getenv("RCC_DENSE_MEMONLY") ? atoi(getenv("RCC_DENSE_MEMONLY")) : 0;
  if (memonly)
    hipLaunchKernelGGL(k_dense_march<1>, dim3(blocks), dim3(256), 0, s, d_grey, w, ht, nstrips, nseg, seg_tiles, nframes,
                       c.thr_min_contrast, c.harris_thresh, c.cand_margin, c.max_candidates, allow_skip, d_bin, d_cand, d_cand_count);
  else
  hipLaunchKernelGGL(k_dense_march<0>, dim3(blocks), dim3(256), 0, s, d_grey, w, ht, nstrips, nseg, seg_tiles, nframes,
                     c.thr_min_contrast, c.harris_thresh, c.cand_margin, c.max_candidates, allow_skip, d_bin, d_cand, d_cand_count);
  return hipGetLastError();
}

// ---- counter calibration: a streaming copy with the dense pass's access width (one dword per lane
// per row) and a known byte count, so that FETCH_SIZE / WRITE_SIZE of k_dense_march can be priced
// (MI355X_MICROARCH.md: those counters are calibrated only for 16 B/lane accesses).
__global__ __launch_bounds__(256) void k_calib_copy_dword(const unsigned* __restrict__ src, unsigned* __restrict__ dst, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
hipError_t rcc_launch_calib_copy(const void* src, void* dst, size_t nbytes, hipStream_t s)
{
  hipLaunchKernelGGL(k_calib_copy_dword, dim3(256 * 16), dim3(256), 0, s, (const unsigned*)src, (unsigned*)dst, nbytes / 4);
  return hipGetLastError();
}
