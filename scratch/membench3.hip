// membench3.hip -- stores-only march: cost as a function of seam placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define W 1920
#define H 1080
typedef unsigned u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// each wave owns pixels [strip*stride, strip*stride+stride) of its rows; lane l covers 4 px at strip*stride + off + 4*l,
// and stores iff l in [l0, l1]
template <int WAVES, int NT>
__global__ __launch_bounds__(64 * WAVES) void st_march(uint8_t* __restrict__ o, int stride, int off, int l0, int l1, int nstrips,
                                                 int nseg, int seg_tiles, int nframes)
{
  const int lane = threadIdx.x & 63;
  const int job = __builtin_amdgcn_readfirstlane(blockIdx.x * WAVES + (threadIdx.x >> 6));
  if (job >= nstrips * nseg * nframes) return;
  const int strip = job % nstrips, seg = (job / nstrips) % nseg, f = job / (nstrips * nseg);
  const int t0 = seg * seg_tiles, t1 = min(t0 + seg_tiles, H / 4);
  const int x0 = strip * stride + off + 4 * lane;
  const bool lo = lane >= l0 && lane <= l1 && x0 >= 0 && x0 < W;
  uint8_t* of = o + (size_t)f * W * H;
  for (int t = t0; t < t1; ++t) {
    if (lo) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        u32* p = reinterpret_cast<u32*>(of + (size_t)(4 * t + k) * W + x0);
        if (NT) __builtin_nontemporal_store((u32)(t + k), p); else *p = (u32)(t + k);
      }
    }
  }
}

template <class F> float timeit(F launch)
{
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  launch(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < 10; ++i) launch();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
  CK(hipGetLastError());
  return ms;
}
int main()
{
  const int NF = 1024;
  const size_t n = (size_t)NF * W * H;
  uint8_t* d; CK(hipMalloc(&d, n));
  CK(hipMemset(d, 0, n));
  const double bytes = (double)n;
  struct Cfg { int stride, off, l0, l1; const char* what; } cfgs[] = {
    {244, -8, 2, 62, "current: seams at 244k"},
    {244, -8, 1, 62, "current + lane 1 (4-B overlap)"},
    {244, -8, 0, 63, "244 stride, all 64 lanes store (12-B overlap)"},
    {256, 0, 0, 63, "aligned 256"},
    {192, 0, 0, 47, "192: seams 64-B aligned"},
    {224, 0, 0, 55, "224: seams 32-B aligned"},
    {240, 0, 0, 59, "240: seams 16-B aligned"},
    {248, 0, 0, 61, "248: seams 8-B aligned"},
    {128, 0, 0, 31, "128"},
    {64, 0, 0, 15, "64"},
  };
  for (int nseg : { 2, 10 }) {
    const int seg_tiles = (270 + nseg - 1) / nseg;
    for (auto c : cfgs) {
      const int nstrips = (W + c.stride - 1) / c.stride;
      const int jobs = nstrips * nseg * NF;
      float m4 = timeit([&] { hipLaunchKernelGGL((st_march<4, 0>), dim3((jobs + 3) / 4), dim3(256), 0, 0, d, c.stride, c.off, c.l0, c.l1, nstrips, nseg, seg_tiles, NF); });
      float m1 = timeit([&] { hipLaunchKernelGGL((st_march<1, 0>), dim3(jobs), dim3(64), 0, 0, d, c.stride, c.off, c.l0, c.l1, nstrips, nseg, seg_tiles, NF); });
      float m8 = timeit([&] { hipLaunchKernelGGL((st_march<8, 0>), dim3((jobs + 7) / 8), dim3(512), 0, 0, d, c.stride, c.off, c.l0, c.l1, nstrips, nseg, seg_tiles, NF); });
      float mn = timeit([&] { hipLaunchKernelGGL((st_march<4, 1>), dim3((jobs + 3) / 4), dim3(256), 0, 0, d, c.stride, c.off, c.l0, c.l1, nstrips, nseg, seg_tiles, NF); });
      printf("nseg %2d %-48s: block4 %.3f ms (%.0f GB/s)  block1 %.3f  block8 %.3f  nontemporal %.3f\n", nseg, c.what, m4, bytes / m4 * 1e-6, m1, m8, mn);
      fflush(stdout);
    }
  }
  return 0;
}
