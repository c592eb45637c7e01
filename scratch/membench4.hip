// membench4.hip -- the data movement of the ingest pass with no arithmetic: read 3 bytes, write 1 byte per pixel
// (1024 x 1080p: 6.2 GB in, 2.1 GB out), as a stream.  Build: hipcc --offload-arch=gfx950 -O3 -o membench4 membench4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// each lane: NU units of 48 B in (three 16-B loads), 16 B out; units of a wave are consecutive (768 B / 256 B contiguous per instruction)
template <int NU, bool NT>
__global__ __launch_bounds__(256) void k(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t nunits)
{
  const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x - (threadIdx.x & 63)) * NU + (threadIdx.x & 63);   // wave-contiguous
  u32x4 a[NU], b[NU], c[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const size_t i = base + (size_t)u * 64;
    if (i < nunits) {
      if (NT) { a[u] = __builtin_nontemporal_load(in + 3 * i); b[u] = __builtin_nontemporal_load(in + 3 * i + 1); c[u] = __builtin_nontemporal_load(in + 3 * i + 2); }
      else { a[u] = in[3 * i]; b[u] = in[3 * i + 1]; c[u] = in[3 * i + 2]; }
    }
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const size_t i = base + (size_t)u * 64;
    if (i < nunits) {
      const u32x4 r = a[u] ^ b[u] ^ c[u];
      if (NT) __builtin_nontemporal_store(r, out + i); else out[i] = r;
    }
  }
}
int main()
{
  const size_t px = (size_t)1920 * 1080 * 1024, nunits = px / 16;
  u32x4 *in, *out;
  if (hipMalloc(&in, px * 3) != hipSuccess || hipMalloc(&out, px) != hipSuccess) return 1;
  (void)hipMemset(in, 1, px * 3);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto run = [&](const char* name, auto kern, int nu) {
    const unsigned blocks = (unsigned)((nunits + 256 * (size_t)nu - 1) / (256 * (size_t)nu));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, in, out, nunits);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 6; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, in, out, nunits);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 6;
    printf("%-28s %.3f ms  %.0f GB/s (4 bytes per pixel)\n", name, ms, 4.0 * px / ms / 1e6); fflush(stdout);
  };
  run("1 unit / lane", k<1, false>, 1);
  run("2 units / lane", k<2, false>, 2);
  run("4 units / lane", k<4, false>, 4);
  run("2 units / lane, nt", k<2, true>, 2);
  run("4 units / lane, nt", k<4, true>, 4);
  return 0;
}
