// k_probe.hip -- measurement aid of include/rcc_debug.h (rcc_debug_measure_clock): what a vector wave-instruction costs
// on THIS device NOW, and the engine clock the chip holds while it issues them.  bench.py's vector-issue roofline of the
// threshold + corner pass used constants for both (2.4 GHz from the device properties, 4.4 cycles from a committed
// microbenchmark of another box); boxes differ by several per cent in the clock they hold under load
// (MI355X_MICROARCH.md, "DVFS give-back" (5)), so the bound is priced with what this launch measures.
//
// The loop is the instruction classes k_dense_wave is made of (dense_rows.h): packed 16-bit add / min / multiply / shift,
// v_dot2, v_perm, DPP moves, three-operand integer adds -- eight classes x eight per iteration, every one a dependent
// update of one of four registers, W waves per SIMD resident on every CU.  Each wave stamps s_memtime (shader cycles)
// and s_memrealtime (100 MHz) around its loop: clock = d(memtime) / d(memrealtime) x 100 MHz; the cost per
// wave-instruction per SIMD comes from HIP events around the launch (ns) -- no assumption about the clock in either.
#include "rcc_internal.h"

#define P8(x) x x x x x x x x
__global__ __launch_bounds__(256) void k_issue_probe(int iters, unsigned long long* __restrict__ stamps, unsigned* __restrict__ sink)
{
  unsigned a = threadIdx.x * 2654435761u, b = a ^ 0x9E3779B9u, c = a + 12345u, d = b + 777u;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    P8(asm volatile("v_pk_add_u16 %0, %1, %0" : "+v"(a) : "v"(b));)
    P8(asm volatile("v_pk_min_u16 %0, %1, %0" : "+v"(b) : "v"(c));)
    P8(asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(d));)
    P8(asm volatile("v_perm_b32 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));)
    P8(asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(c));)
    P8(asm volatile("v_pk_mul_lo_u16 %0, %1, %0" : "+v"(b) : "v"(d));)
    P8(asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));)
    P8(asm volatile("v_pk_ashrrev_i16 %0, 3, %1" : "=v"(d) : "v"(c));)
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) {
    const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    stamps[2 * w] = t1 - t0;
    stamps[2 * w + 1] = r1 - r0;
  }
  if (a + b + c + d == 0x12345678u) sink[0] = a;      // keeps the chain live; practically never taken
}

hipError_t rcc_launch_issue_probe(int blocks, int iters, unsigned long long* d_stamps, unsigned* d_sink, hipStream_t s)
{
  hipLaunchKernelGGL(k_issue_probe, dim3(blocks), dim3(256), 0, s, iters, d_stamps, d_sink);
  return hipGetLastError();
}
