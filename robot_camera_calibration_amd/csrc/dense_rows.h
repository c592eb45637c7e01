// dense_rows.h -- the per-lane row pipeline of the threshold + corner pass (a3 + a4.1), shared by the two
// row-marching kernels (k_dense_fast.hip: one wave per strip, registers only; k_dense_band.hip: one workgroup
// per full-width band, grey rows staged through LDS).  Definitions: DESIGN.md section 3; bit-exact with
// k_dense_lds (k_dense.hip) and with the CPU restatement the tests check against.
//
// Lane layout: a wavefront covers a 256-pixel window, lane l holds the 4 pixels x0 .. x0+3 (one dword of the
// grey row); lanes 0, 1 and 63 are halo (the left lattice neighbour of the first useful pixel needs 5 pixels),
// so a window has 244 useful pixels.  Neighbouring lanes exchange edge values with whole-wave DPP shifts.
#pragma once
#include "rcc_internal.h"

typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

#define DPP_FROM_LEFT 0x138   // wave_shr:1 : lane l reads lane l-1
#define DPP_FROM_RIGHT 0x130  // wave_shl:1 : lane l reads lane l+1
#define STRIP_USE 244         // useful pixels per window: lanes 2..62

__device__ __forceinline__ int from_left(int v, int edge) { return __builtin_amdgcn_update_dpp(edge, v, DPP_FROM_LEFT, 0xf, 0xf, false); }
__device__ __forceinline__ int from_right(int v, int edge) { return __builtin_amdgcn_update_dpp(edge, v, DPP_FROM_RIGHT, 0xf, 0xf, false); }
// zero for lanes without a source (bound_ctrl:0): lets the compiler fold the move into the consumer
__device__ __forceinline__ int from_left0(int v) { return __builtin_amdgcn_update_dpp(0, v, DPP_FROM_LEFT, 0xf, 0xf, true); }
__device__ __forceinline__ int from_right0(int v) { return __builtin_amdgcn_update_dpp(0, v, DPP_FROM_RIGHT, 0xf, 0xf, true); }
// "does any lane hold c": the ballot compared as a scalar (hipcc's __any turns a lane mask into 0 / 1 per lane and compares
// that: two vector instructions)
__device__ __forceinline__ bool wave_any(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0ull; }
__device__ __forceinline__ i16x2 as_i(unsigned v) { return __builtin_bit_cast(i16x2, v); }
__device__ __forceinline__ u16x2 as_u(unsigned v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ unsigned bits(i16x2 v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ unsigned bits(u16x2 v) { return __builtin_bit_cast(unsigned, v); }

// the same two moves pinned to where they are written (the compiler hoists the builtin form out of a rarely taken
// branch and pays for it on every row); s_nop 1: the wait states a DPP read needs after a vector write of its source
__device__ __forceinline__ int from_left0_here(int v)
{
  int d;
  asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(d) : "v"(v));
  return d;
}
__device__ __forceinline__ int from_right0_here(int v)
{
  int d;
  asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(d) : "v"(v));
  return d;
}
// lane l + lane l-1's / lane l+1's value in ONE instruction (VOP2 with a DPP source; lanes without a source add 0): the
// compiler folds a DPP move into a two-operand add but not into v_add3
__device__ __forceinline__ int add_from_left0(int from, int addend) { return addend + from_left0(from); }
__device__ __forceinline__ int add_from_right0(int from, int addend) { return addend + from_right0(from); }
// dot2 with a zero accumulator in its VOP3P form (the builtin becomes v_mov 0 + v_dot2c)
__device__ __forceinline__ int dot2z(i16x2 a, i16x2 b)
{
  int d;
  asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

struct HSum { int xx0, xy0, yy0, xx2, xy2, yy2; };   // raw 5-px row sums at pixel 0 and pixel 2 of the lane
struct SobelRow { i16x2 dh01, dh23, sh01, sh23; };     // horizontal Sobel partials of one row (pixel pairs 0-1, 2-3)
struct Tile4 { unsigned g0, g1, g2, g3; };             // grey dwords of the four rows of a tile row
struct TStat { int hmin, hmax; };                      // horizontally dilated tile min / max
struct LRow { int r0, r2; };                           // lattice row of responses at the lane's px 0 and px 2 (the neighbour lanes'
                                                       // values are fetched by DPP only where a row is searched for maxima)

// "don't care" values that cost no instruction (the compiler may leave anything in the register)
__device__ __forceinline__ void dontcare(int& v) { asm volatile("" : "=v"(v)); }
__device__ __forceinline__ void dontcare(SobelRow& q)
{
  int a, b, c, d; dontcare(a); dontcare(b); dontcare(c); dontcare(d);
  q.dh01 = __builtin_bit_cast(i16x2, a); q.dh23 = __builtin_bit_cast(i16x2, b); q.sh01 = __builtin_bit_cast(i16x2, c); q.sh23 = __builtin_bit_cast(i16x2, d);
}
__device__ __forceinline__ void dontcare(HSum& q) { dontcare(q.xx0); dontcare(q.xy0); dontcare(q.yy0); dontcare(q.xx2); dontcare(q.xy2); dontcare(q.yy2); }

// horizontally dilated min / max of the lane's 4x4 tile (rows C.g0..g3)
__device__ __forceinline__ TStat tile_stats(const Tile4& C)
{
  u16x2 tmn = (u16x2)(255), tmx = (u16x2)(0);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned G = (k == 0) ? C.g0 : (k == 1) ? C.g1 : (k == 2) ? C.g2 : C.g3;
    const u16x2 n0 = as_u(__builtin_amdgcn_perm(0u, G, 0x0C010C00u));
    const u16x2 n1 = as_u(__builtin_amdgcn_perm(0u, G, 0x0C030C02u));
    tmn = __builtin_elementwise_min(tmn, __builtin_elementwise_min(n0, n1));
    tmx = __builtin_elementwise_max(tmx, __builtin_elementwise_max(n0, n1));
  }
  const u16x2 nmn = (u16x2)(255) - tmn;                 // complemented while still packed: one instruction for both halves
  const int nmin = max((int)nmn.x, (int)nmn.y), tmax = max((int)tmx.x, (int)tmx.y);
  // neighbours by DPP with zero fill for the lanes that have none: harmless for a maximum of non-negative values, so
  // the minimum is taken as the maximum of 255 - v.  Each max folds its DPP source (v_max_i32_dpp): 2 + 4 instructions
  // instead of 4 copies + 4 DPP moves + min3 / max3.
  TStat hn;
  int m1 = max(tmax, from_left0(tmax)), n1 = max(nmin, from_left0(nmin));
  asm volatile("" : "+v"(m1), "+v"(n1));       // keeps the two maxima apart: as v_max3 the DPP sources become two moves
  hn.hmax = max(m1, from_right0(tmax));
  hn.hmin = 255 - max(n1, from_right0(nmin));
  return hn;
}

// threshold of four pixels: per byte 255 if v > thr else 0; 127 everywhere if the tile is flat.
// v > thr <=> v + (0x8000 - (thr + 1)) >= 0x8000 in a 16-bit field (thr <= 254 whenever the tile is not flat, so the
// field cannot overflow): the pixels are spread over two dwords of 16-bit fields (even bytes by a mask, odd bytes by
// a byte permute), the constant is added to both, and ONE v_perm_b32 with sign-replicating selectors (8..11: bit 15 /
// 31 of either source as 0x00 / 0xFF) puts the four results back in pixel order.  6 instructions per dword with the
// flat select (the carry-free byte compare it replaces took 9).
struct Thr4 {
  unsigned k; bool flat;
  __device__ __forceinline__ Thr4(int thr, bool flat_) : flat(flat_) { k = __umul24((unsigned)(0x7FFF - thr), 0x10001u); }
  __device__ __forceinline__ unsigned operator()(unsigned x) const
  {
    const unsigned e = (x & 0x00FF00FFu) + k;                               // [p0, p2]
    const unsigned o = __builtin_amdgcn_perm(0u, x, 0x0C030C01u) + k;      // [p1, p3]
    // {S0 = e, S1 = o}: selector 10 / 11 = sign of e's low / high half, 8 / 9 = of o's
    const unsigned r = __builtin_amdgcn_perm(e, o, 0x090B080Au);
    return flat ? 0x7F7F7F7Fu : r;
  }
};

// the Sobel / structure-tensor / response / selection pipeline of one lane
struct RowPipe {
  HSum hprev, qa, qb;          // qa: pair closed at k=0, qb: pair closed at k=2
  LRow Ra, Rb;                 // lattice rows y-4, y-2
  i16x2 two_splat;             // (2, 2), opaque to the optimiser: see row()
  // job constants
  // xlane0 = first pixel of lane 0 (wave-uniform; the lane's own column and index are re-derived where a candidate is
  // written instead of living in two registers through the loop); ok0 / ok2: may this lane report a candidate at its
  // px 0 / px 2 (useful lane, column inside the window's output range and the margin) -- loop-invariant lane masks
  int xlane0, w, h, t0, t1, margin, hthresh, cap, f;
  bool ok0, ok2;
  rcc_cand* cand; int32_t* cand_count;

  // (after w and margin are set)  x0: the lane's first pixel; lane_out: useful lane whose pixels lie in the columns this window reports
  __device__ __forceinline__ void set_lane(int x0, int lane, bool lane_out)
  {
    xlane0 = __builtin_amdgcn_readfirstlane(x0 - 4 * lane);
    ok0 = lane_out && x0 >= margin && x0 < w - margin;
    ok2 = lane_out && x0 + 2 >= margin && x0 + 2 < w - margin;
  }
  __device__ __forceinline__ void reset()
  {
    unsigned t2 = 0x00020002u;
    asm volatile("" : "+v"(t2));
    two_splat = __builtin_bit_cast(i16x2, t2);
    hprev = HSum{ 0, 0, 0, 0, 0, 0 }; qa = hprev; qb = hprev;
    Ra = LRow{ INT32_MIN, INT32_MIN }; Rb = Ra;
  }
  // a skipped tile row: the two lattice rows not produced lie in flat tiles; the row sums now describe rows
  // that were not read, and everything they can still reach is a masked response, so their contents do not
  // matter (no copies at the control-flow join)
  __device__ __forceinline__ void skip()
  {
    Ra.r0 = Ra.r2 = INT32_MIN; Rb = Ra;
    dontcare(hprev); dontcare(qa); dontcare(qb);
  }

  // one image row: G = its grey dword; (a, b) = Sobel partials of rows r-2, r-1; n receives row r's.
  // rmask != 0: the lane's tile at the lattice row being produced is flat => its response cannot
  // reach hthresh (host-checked bound), so it is replaced by INT32_MIN
  __device__ __forceinline__ void row(const int r, const int k, const unsigned G, const SobelRow& a, const SobelRow& b, SobelRow& n, const bool rmask)
  {
    // ---- stage A (row r): horizontal Sobel partials, natural-order pixel pairs
    // zero for the lanes without a neighbour: that only changes lane 0's pixel -1 and lane 63's pixel 4, which
    // reach nothing outside those (halo) lanes' own sums
    const unsigned GL = (unsigned)from_left0((int)G), GR = (unsigned)from_right0((int)G);
    const u16x2 n0 = as_u(__builtin_amdgcn_perm(0u, G, 0x0C010C00u));    // [p0,p1]
    const u16x2 n1 = as_u(__builtin_amdgcn_perm(0u, G, 0x0C030C02u));    // [p2,p3]
    const u16x2 mm = as_u(__builtin_amdgcn_perm(0u, G, 0x0C020C01u));    // [p1,p2]
    const u16x2 lh = as_u(__builtin_amdgcn_perm(GL, G, 0x0C000C07u));    // [p-1,p0]
    const u16x2 rh = as_u(__builtin_amdgcn_perm(GR, G, 0x0C040C03u));    // [p3,p4]
    // 2 * x + y as ONE v_pk_mad_i16: the multiplier comes from a register the optimiser cannot see through (a literal 2
    // is strength-reduced to a shift and the fused form is lost)
    const i16x2 two = two_splat;
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
    //   sum at px 0 = left lane's (p2, p3) + own (p0, p1) + own p2;  sum at px 2 = own (p0 .. p3) + right lane's p0.
    // Seven instructions per quantity: the single products ride in as dot2 accumulators, the neighbour terms as DPP
    // sources of two-operand adds (ten with the builtin dot2 on a zero accumulator and three-operand adds).
    const int ax0 = gx01.x, ay0 = gy01.x, ax2 = gx23.x, ay2 = gy23.x;
    const int q0xx = __mul24(ax0, ax0), q0xy = __mul24(ax0, ay0), q0yy = __mul24(ay0, ay0);
    const int q2xx = __mul24(ax2, ax2), q2xy = __mul24(ax2, ay2), q2yy = __mul24(ay2, ay2);
    const int d1xx = dot2z(gx23, gx23), d1xy = dot2z(gx23, gy23), d1yy = dot2z(gy23, gy23);
    HSum hc;
    hc.xx0 = add_from_left0(d1xx, __builtin_amdgcn_sdot2(gx01, gx01, q2xx, false));
    hc.xy0 = add_from_left0(d1xy, __builtin_amdgcn_sdot2(gx01, gy01, q2xy, false));
    hc.yy0 = add_from_left0(d1yy, __builtin_amdgcn_sdot2(gy01, gy01, q2yy, false));
    hc.xx2 = __builtin_amdgcn_sdot2(gx01, gx01, add_from_right0(q0xx, d1xx), false);
    hc.xy2 = __builtin_amdgcn_sdot2(gx01, gy01, add_from_right0(q0xy, d1xy), false);
    hc.yy2 = __builtin_amdgcn_sdot2(gy01, gy01, add_from_right0(q0yy, d1yy), false);
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
      hprev = hc;
      // selection on lattice row yc = rho - 4 = r - 5 (rows Ra = yc-2, Rb = yc, Rn = yc+2)
      const int yc = r - 5;
      const bool rowok = (yc >= 4 * t0) && (yc < 4 * t1) && (yc >= margin) && (yc < h - margin);
      if (rowok && wave_any((Rb.r0 >= hthresh) || (Rb.r2 >= hthresh))) {
        // the neighbour lanes' responses, here only (the branch is wave-uniform: every lane is live for the DPP moves);
        // lanes 0 / 63 (halo, never selected) see 0 instead of a neighbour
        const int RaL = from_left0_here(Ra.r2), RbL = from_left0_here(Rb.r2), RnL = from_left0_here(Rn.r2);
        const int RaR = from_right0_here(Ra.r0), RbR = from_right0_here(Rb.r0), RnR = from_right0_here(Rn.r0);
        bool is0 = ok0 && Rb.r0 >= hthresh &&
                   Rb.r0 > RaL && Rb.r0 > Ra.r0 && Rb.r0 > Ra.r2 && Rb.r0 > RbL &&
                   Rb.r0 >= Rb.r2 && Rb.r0 >= RnL && Rb.r0 >= Rn.r0 && Rb.r0 >= Rn.r2;
        bool is2 = ok2 && Rb.r2 >= hthresh &&
                   Rb.r2 > Ra.r0 && Rb.r2 > Ra.r2 && Rb.r2 > RaR && Rb.r2 > Rb.r0 &&
                   Rb.r2 >= RbR && Rb.r2 >= Rn.r0 && Rb.r2 >= Rn.r2 && Rb.r2 >= RnR;
        const unsigned long long m0 = __ballot(is0), m2 = __ballot(is2);
        const int n0c = __popcll(m0), n2c = __popcll(m2);
        if (n0c + n2c) {
          int lane;
          asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
          const int xa = xlane0 + 4 * lane, xb = xa + 2;
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
  }
};
