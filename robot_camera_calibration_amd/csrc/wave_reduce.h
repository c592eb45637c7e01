// wave_reduce.h -- sums over the 64 lanes of a wavefront (fp64), gfx950.
//
// halve<B>(lane, a, b): one exchange+add at lane bit B -- lanes with bit B clear return a + their partner's a,
// the others b + their partner's b.  With a == b it is one step of an all-reduce (both partners end with the
// same sum, own + partner's: IEEE addition commutes, so the pairing tree bit 5, 4, .., 0 rounds exactly like the
// xor butterfly v += shfl_xor(v, 32 >> k)).  With a != b it is one step of a recursive-halving reduce-scatter:
// N quantities cost P-1 exchange+add steps (P = N rounded up to a power of two) instead of 6 N.
// Exchanges: v_permlane32_swap / v_permlane16_swap (gfx950) for lane bits 5 and 4, DPP row rotations with bank
// masks for bits 3 and 2, DPP quad permutes for bits 1 and 0 (lane maps checked by scratch/t_swap.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace wred {
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double mk(unsigned lo, unsigned hi) { return __builtin_bit_cast(double, (unsigned long long)lo | ((unsigned long long)hi << 32)); }
__device__ __forceinline__ unsigned lo32(double d) { return (unsigned)__builtin_bit_cast(unsigned long long, d); }
__device__ __forceinline__ unsigned hi32(double d) { return (unsigned)(__builtin_bit_cast(unsigned long long, d) >> 32); }
template <int CTRL, int BANKS>
__device__ __forceinline__ double dpp_merge(double old, double v)
{
  return mk(__builtin_amdgcn_update_dpp(lo32(old), lo32(v), CTRL, 0xf, BANKS, false),
            __builtin_amdgcn_update_dpp(hi32(old), hi32(v), CTRL, 0xf, BANKS, false));
}
template <int B>
__device__ __forceinline__ double halve(int lane, double a, double b)
{
  if (B == 5) {
    const u32x2_t l = __builtin_amdgcn_permlane32_swap(lo32(a), lo32(b), false, false);
    const u32x2_t h = __builtin_amdgcn_permlane32_swap(hi32(a), hi32(b), false, false);
    return mk(l.x, h.x) + mk(l.y, h.y);
  } else if (B == 4) {
    const u32x2_t l = __builtin_amdgcn_permlane16_swap(lo32(a), lo32(b), false, false);
    const u32x2_t h = __builtin_amdgcn_permlane16_swap(hi32(a), hi32(b), false, false);
    return mk(l.x, h.x) + mk(l.y, h.y);
  } else if (B == 3) {
    const double recv = dpp_merge<0x128, 0x3>(dpp_merge<0x128, 0xC>(0.0, b), a);      // row_ror:8
    return ((lane & 8) ? b : a) + recv;
  } else if (B == 2) {
    const double recv = dpp_merge<0x12C, 0x5>(dpp_merge<0x124, 0xA>(0.0, b), a);      // row_ror:12 / row_ror:4
    return ((lane & 4) ? b : a) + recv;
  } else {
    const bool up = (lane & (1 << B)) != 0;
    const double send = up ? a : b, keep = up ? b : a;
    const double recv = (B == 1) ? dpp_merge<0x4E, 0xF>(0.0, send) : dpp_merge<0xB1, 0xF>(0.0, send);   // quad_perm [2,3,0,1] / [1,0,3,2]
    return keep + recv;
  }
}
// P quantities remain per lane before the level of lane bit B; P/2 after (P >= 2), or the same one summed (P == 1)
template <int P, int B>
__device__ __forceinline__ void level(int lane, double* v)
{
  if (P >= 2) {
#pragma unroll
    for (int j = 0; j < P / 2; ++j) v[j] = halve<B>(lane, v[j], v[j + P / 2]);
  } else {
    v[0] = halve<B>(lane, v[0], v[0]);
  }
}
// all six levels: afterwards v[0] of lane l holds the total of quantity l / (64 / P)
template <int P>
__device__ __forceinline__ void reduce_scatter(int lane, double* v)
{
  level<P, 5>(lane, v);
  level<(P >= 2 ? P / 2 : 1), 4>(lane, v);
  level<(P >= 4 ? P / 4 : 1), 3>(lane, v);
  level<(P >= 8 ? P / 8 : 1), 2>(lane, v);
  level<(P >= 16 ? P / 16 : 1), 1>(lane, v);
  level<(P >= 32 ? P / 32 : 1), 0>(lane, v);
}
__device__ __forceinline__ double all_sum(int lane, double v)
{
  reduce_scatter<1>(lane, &v);
  return v;
}
// ---- all-reduce with any commutative, associative operation on 32- or 64-bit values: own (x) and the
// partner's (y) value at lane bit B, without ds_bpermute
template <int B>
__device__ __forceinline__ void pair32(unsigned v, unsigned& x, unsigned& y)
{
  if (B == 5) { const u32x2_t r = __builtin_amdgcn_permlane32_swap(v, v, false, false); x = r.x; y = r.y; }
  else if (B == 4) { const u32x2_t r = __builtin_amdgcn_permlane16_swap(v, v, false, false); x = r.x; y = r.y; }
  else if (B == 3) { x = v; y = __builtin_amdgcn_update_dpp(0u, v, 0x128, 0xf, 0xf, false); }                       // row_ror:8
  else if (B == 2) { x = v; y = __builtin_amdgcn_update_dpp(__builtin_amdgcn_update_dpp(0u, v, 0x124, 0xf, 0xA, false), v, 0x12C, 0xf, 0x5, false); }
  else if (B == 1) { x = v; y = __builtin_amdgcn_update_dpp(0u, v, 0x4E, 0xf, 0xf, false); }                         // quad_perm [2,3,0,1]
  else { x = v; y = __builtin_amdgcn_update_dpp(0u, v, 0xB1, 0xf, 0xf, false); }                                      // quad_perm [1,0,3,2]
}
template <class Op>
__device__ __forceinline__ unsigned all_reduce32(unsigned v, Op op)
{
  unsigned x, y;
  pair32<5>(v, x, y); v = op(x, y);
  pair32<4>(v, x, y); v = op(x, y);
  pair32<3>(v, x, y); v = op(x, y);
  pair32<2>(v, x, y); v = op(x, y);
  pair32<1>(v, x, y); v = op(x, y);
  pair32<0>(v, x, y); v = op(x, y);
  return v;
}
template <class Op>
__device__ __forceinline__ unsigned long long all_reduce64(unsigned long long v, Op op)
{
#define WRED_STEP64(B)                                                                                              \
  {                                                                                                                 \
    unsigned xl, yl, xh, yh;                                                                                        \
    pair32<B>((unsigned)v, xl, yl);                                                                                 \
    pair32<B>((unsigned)(v >> 32), xh, yh);                                                                         \
    v = op((unsigned long long)xl | ((unsigned long long)xh << 32), (unsigned long long)yl | ((unsigned long long)yh << 32)); \
  }
  WRED_STEP64(5) WRED_STEP64(4) WRED_STEP64(3) WRED_STEP64(2) WRED_STEP64(1) WRED_STEP64(0)
#undef WRED_STEP64
  return v;
}

// value of lane `src` (compile-time) in every lane, through scalar registers
template <int SRC>
__device__ __forceinline__ double bcast(double v)
{
  return mk((unsigned)__builtin_amdgcn_readlane((int)lo32(v), SRC), (unsigned)__builtin_amdgcn_readlane((int)hi32(v), SRC));
}
}  // namespace wred
