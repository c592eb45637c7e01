// vbench.hip -- issue cost of the vector instructions the threshold + corner pass is made of (gfx950), measured as
// cycles per wave-instruction per SIMD with W waves per SIMD resident on every CU.  Build: hipcc --offload-arch=gfx950 -O3 -o vbench vbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP16(x) x x x x x x x x x x x x x x x x
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
template <int OP>
__global__ __launch_bounds__(256) void k(int iters, unsigned* out)
{
  unsigned a = threadIdx.x * 2654435761u, b = a ^ 0x9E3779B9u, c = a + 12345u, d = b + 777u;
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) { REP64(asm volatile("v_add_u32 %0, %1, %0" : "+v"(a) : "v"(b));) }
    if (OP == 1) { REP64(asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b));) }
    if (OP == 2) { REP64(asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b));) }
    if (OP == 3) { REP64(asm volatile("v_pk_add_u16 %0, %1, %0" : "+v"(a) : "v"(b));) }
    if (OP == 4) { REP64(asm volatile("v_dot2c_i32_i16 %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
    if (OP == 5) { REP64(asm volatile("v_mul_i32_i24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0" : "=v"(a) : "v"(b), "v"(c));) }
    if (OP == 6) { REP64(asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a) : "v"(b), "v"(c), "v"(d));) }
    if (OP == 7) { REP64(asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));) }
    if (OP == 8) { REP64(asm volatile("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b));) }
    if (OP == 9) { REP64(asm volatile("v_add_u32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b));) }
    if (OP == 10) { REP64(asm volatile("v_pk_ashrrev_i16 %0, 3, %1" : "=v"(a) : "v"(b));) }
    if (OP == 11) { REP64(asm volatile("v_pk_mad_i16 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));) }
    if (OP == 12) { REP64(asm volatile("v_mul_i32_i24 %0, %1, %2" : "=v"(a) : "v"(b), "v"(c));) }
    if (OP == 13) { REP64(asm volatile("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(a) : "v"(b), "v"(c));) }
    if (OP == 14) { REP64(asm volatile("v_dot4_i32_i8 %0, %1, %2, 0" : "=v"(a) : "v"(b), "v"(c));) }
    if (OP == 15) { REP64(asm volatile("v_mov_b32_dpp %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(a) : "v"(b));) }
    if (OP == 16) { REP64(asm volatile("v_pk_sub_i16 %0, %1, %0" : "+v"(a) : "v"(b));) }
    if (OP == 17) { REP64(asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));) }
    if (OP == 18) { REP64(asm volatile("v_pk_min_u16 %0, %1, %0" : "+v"(a) : "v"(b));) }
    if (OP == 19) { REP64(asm volatile("v_alignbyte_b32 %0, %1, %2, 1" : "=v"(a) : "v"(b), "v"(c));) }
    if (OP == 20) { REP64(asm volatile("v_pk_mul_lo_u16 %0, %1, %0" : "+v"(a) : "v"(b));) }
    if (OP == 21) { REP64(asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a) : "v"(b), "v"(c));) }
    if (OP == 22) { REP64(asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b));) }
    if (OP == 23) { REP64(asm volatile("ds_swizzle_b32 %0, %1 offset:swizzle(SWAP,1)\n\ts_waitcnt lgkmcnt(0)" : "=v"(a) : "v"(b));) }
    if (OP == 24) { REP64(asm volatile("v_min3_i32 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));) }
    if (OP == 25) { REP64(asm volatile("v_bfe_i32 %0, %1, 16, 16" : "=v"(a) : "v"(b));) }
    if (OP == 26) { REP64(asm volatile("v_and_b32 %0, %1, %0" : "+v"(a) : "v"(b));) }
    if (OP == 27) { REP64(asm volatile("v_lshrrev_b32 %0, 3, %1" : "=v"(a) : "v"(b));) }
    if (OP == 28) { REP64(asm volatile("v_max_i32 %0, %1, %0" : "+v"(a) : "v"(b));) }
    if (OP == 29) { REP64(asm volatile("v_mov_b32 %0, %1" : "=v"(a) : "v"(b));) }
    if (OP == 30) { REP64(asm volatile("v_sub_u32 %0, %1, %0" : "+v"(a) : "v"(b));) }
    if (OP == 31) { REP64(asm volatile("v_cndmask_b32_e64 %0, %1, %2, s[10:11]" : "=v"(a) : "v"(b), "v"(c) : "s10", "s11");) }
    if (OP == 32) { REP64(asm volatile("v_cmp_gt_i32 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");) }
    if (OP == 33) { REP64(asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(a) : "v"(b), "v"(c));) }
    if (OP == 34) { REP64(asm volatile("v_max_u16 %0, %1, %0" : "+v"(a) : "v"(b));) }
    if (OP == 35) { REP64(asm volatile("v_add_u16 %0, %1, %0" : "+v"(a) : "v"(b));) }
    if (OP == 36) { REP64(asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));) }
    if (OP == 37) { REP64(asm volatile("v_lshl_add_u32 %0, %1, 2, %0" : "+v"(a) : "v"(b));) }
    if (OP == 38) { REP64(asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));) }
    if (OP == 39) { REP64(asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
    if (OP == 40) { REP64(asm volatile("v_add_f32 %0, %1, %0" : "+v"(a) : "v"(b));) }
    if (OP == 41) { REP64(asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(a) : "v"(b));) }
  }
  if (a == 0x12345678u) out[0] = a + b + c + d;
}
static const char* names[] = { "v_add_u32", "v_mov_dpp wave_shr:1", "v_mov_dpp row_shr:1", "v_pk_add_u16", "v_dot2c_i32_i16", "v_mul_i32_i24_sdwa",
  "v_perm_b32", "v_add3_u32", "v_mov_dpp wave_shl:1", "v_add_u32_dpp wave_shr:1", "v_pk_ashrrev_i16", "v_pk_mad_i16", "v_mul_i32_i24", "v_dot2_i32_i16 (vop3p)",
  "v_dot4_i32_i8", "v_mov_dpp row_bcast:15", "v_pk_sub_i16", "v_mad_i32_i24", "v_pk_min_u16", "v_alignbyte_b32", "v_pk_mul_lo_u16", "v_cndmask_b32", "v_mov_dpp quad_perm",
  "ds_swizzle+wait", "v_min3_i32", "v_bfe_i32", "v_and_b32", "v_lshrrev_b32", "v_max_i32", "v_mov_b32", "v_sub_u32", "v_cndmask_b32_e64 (sgpr mask)",
  "v_cmp_gt_i32 vcc", "v_mul_u32_u24", "v_max_u16", "v_add_u16", "v_sad_u8", "v_lshl_add_u32", "v_fma_f32", "v_fmac_f32", "v_add_f32", "v_add_u32_sdwa" };
typedef void (*kern_t)(int, unsigned*);
template <int N> struct Tab { static void fill(kern_t* t) { t[N] = k<N>; Tab<N - 1>::fill(t); } };
template <> struct Tab<-1> { static void fill(kern_t*) {} };
int main()
{
  const int NOPS = 42;
  kern_t tab[NOPS]; Tab<NOPS - 1>::fill(tab);
  unsigned* out; hipMalloc(&out, 4);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  for (int wps : { 1, 4, 8 }) {
    printf("--- %d wave(s) per SIMD, %d CUs, clock from device props %.0f MHz\n", wps, cus, p.clockRate / 1000.0);
    for (int op = 0; op < NOPS; ++op) {
      const int blocks = cus * wps;                 // 256 threads = 4 waves = one per SIMD
      hipLaunchKernelGGL(tab[op], dim3(blocks), dim3(256), 0, 0, 10, out);
      hipEventRecord(e0);
      hipLaunchKernelGGL(tab[op], dim3(blocks), dim3(256), 0, 0, iters, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double instr_per_simd = (double)iters * 64 * wps;
      printf("%-28s %8.3f ms  %6.2f ns per wave-instr per SIMD = %5.2f cycles at 2.4 GHz\n", names[op], ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    }
  }
  return 0;
}
