// vbench_ilp.hip -- does ONE wave per SIMD issue independent vector instructions faster than dependent ones?  (gfx950)
// Build: hipcc --offload-arch=gfx950 -O3 -o vbench_ilp vbench_ilp.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ __launch_bounds__(256) void k(int iters, unsigned* out)
{
  unsigned a = threadIdx.x * 2654435761u, b = a ^ 0x9E3779B9u, c = a + 12345u, d = b + 777u, e = a + 5u, f = b + 9u, g = c + 1u, h = d + 3u, x = a | 1u;
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) { REP16(asm volatile("v_pk_add_u16 %0, %1, %0\n\tv_pk_add_u16 %0, %1, %0\n\tv_pk_add_u16 %0, %1, %0\n\tv_pk_add_u16 %0, %1, %0" : "+v"(a) : "v"(x));) }                       // one chain
    if (OP == 1) { REP16(asm volatile("v_pk_add_u16 %0, %4, %0\n\tv_pk_add_u16 %1, %4, %1\n\tv_pk_add_u16 %2, %4, %2\n\tv_pk_add_u16 %3, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x));) }   // four chains
    if (OP == 2) { REP16(asm volatile("v_pk_add_u16 %0, %2, %0\n\tv_pk_add_u16 %1, %2, %1\n\tv_pk_add_u16 %0, %2, %0\n\tv_pk_add_u16 %1, %2, %1" : "+v"(a), "+v"(b) : "v"(x));) }                     // two chains
    if (OP == 3) { REP16(asm volatile("v_add_u32 %0, %1, %0\n\tv_add_u32 %0, %1, %0\n\tv_add_u32 %0, %1, %0\n\tv_add_u32 %0, %1, %0" : "+v"(a) : "v"(x));) }
    if (OP == 4) { REP16(asm volatile("v_add_u32 %0, %4, %0\n\tv_add_u32 %1, %4, %1\n\tv_add_u32 %2, %4, %2\n\tv_add_u32 %3, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x));) }
    if (OP == 5) { REP16(asm volatile("v_pk_add_u16 %0, %4, %0\n\ts_add_u32 s20, s20, 1\n\tv_pk_add_u16 %1, %4, %1\n\ts_add_u32 s21, s21, 1\n\tv_pk_add_u16 %2, %4, %2\n\ts_add_u32 s22, s22, 1\n\tv_pk_add_u16 %3, %4, %3\n\ts_add_u32 s23, s23, 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x) : "s20", "s21", "s22", "s23", "scc");) }   // four chains + 4 scalar
    if (OP == 6) { REP16(asm volatile("v_mov_b32_dpp %0, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %2, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x));) }
    if (OP == 7) { REP16(asm volatile("v_dot2c_i32_i16 %0, %4, %4\n\tv_dot2c_i32_i16 %1, %4, %4\n\tv_dot2c_i32_i16 %2, %4, %4\n\tv_dot2c_i32_i16 %3, %4, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x));) }
  }
  if (a == 0x12345678u) out[0] = a + b + c + d + e + f + g + h;
}
static const char* names[] = { "pk_add, one chain", "pk_add, four chains", "pk_add, two chains", "v_add_u32, one chain", "v_add_u32, four chains", "pk_add four chains + 4 s_add", "dpp mov wave_shr four chains", "dot2c four chains" };
typedef void (*kern_t)(int, unsigned*);
int main()
{
  kern_t tab[8] = { k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7> };
  unsigned* out; (void)hipMalloc(&out, 4);
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 2000;
  for (int wps : { 1, 2, 3, 6 }) {
    printf("--- %d wave(s) per SIMD\n", wps);
    for (int op = 0; op < 8; ++op) {
      hipLaunchKernelGGL(tab[op], dim3(cus * wps), dim3(256), 0, 0, 10, out);
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(tab[op], dim3(cus * wps), dim3(256), 0, 0, iters, out);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      const double ninst = (double)iters * 64 * (op == 5 ? 1 : 1);     // vector instructions per wave
      printf("%-32s %8.3f ms  %6.2f ns per vector instr per wave = %5.2f cycles at 2.4 GHz; per SIMD %5.2f cycles\n", names[op], ms, ms * 1e6 / ninst, ms * 1e6 / ninst * 2.4, ms * 1e6 / ninst * 2.4 / wps);
      fflush(stdout);
    }
  }
  return 0;
}
