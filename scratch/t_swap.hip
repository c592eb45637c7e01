// experiment: semantics of v_permlane32_swap / v_permlane16_swap and DPP row_ror with bank masks on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out)
{
  const unsigned lane = threadIdx.x;
  unsigned a = 100 + lane, b = 200 + lane;
  u2 r32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  u2 r16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[lane] = r32.x; out[64 + lane] = r32.y; out[128 + lane] = r16.x; out[192 + lane] = r16.y;
  // xor 8 inside a row: row_ror:8 = 0x128
  out[256 + lane] = __builtin_amdgcn_update_dpp(0u, a, 0x128, 0xf, 0xf, false);
  // row_ror:4 (0x124) and row_ror:12 (0x12C)
  out[320 + lane] = __builtin_amdgcn_update_dpp(0u, a, 0x124, 0xf, 0xf, false);
  out[384 + lane] = __builtin_amdgcn_update_dpp(0u, a, 0x12C, 0xf, 0xf, false);
  // bank mask 0x3 with old = 999
  out[448 + lane] = __builtin_amdgcn_update_dpp(999u, a, 0x128, 0xf, 0x3, false);
}
int main()
{
  unsigned* d; hipMalloc(&d, 512 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[8] = { "swap32.x(a)", "swap32.y(b)", "swap16.x(a)", "swap16.y(b)", "row_ror8(a)", "row_ror4(a)", "row_ror12(a)", "ror8 bank3 old999" };
  for (int r = 0; r < 8; ++r) { printf("%-18s:", names[r]); for (int i = 0; i < 64; ++i) printf(" %u", h[r * 64 + i]); printf("\n"); }
  return 0;
}
