// membench2.hip -- loads-only march: cost as a function of strip stride / offset / width
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define W 1920
#define H 1080
typedef unsigned u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void rd_stream(const u32x4* __restrict__ s, u32* __restrict__ d, size_t n)
{
  u32x4 a = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a ^= s[i];
  if ((a.x ^ a.y ^ a.z ^ a.w) == 0x12345u) d[0] = 1;
}

template <int DEPTH, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void rd_march(const uint8_t* __restrict__ g, uint8_t* __restrict__ o, int stride, int off, int nstrips,
                                                 int nseg, int seg_tiles, int nframes, int rowstride, int framestride)
{
  const int lane = threadIdx.x & 63;
  const int job = __builtin_amdgcn_readfirstlane(blockIdx.x * WAVES + (threadIdx.x >> 6));
  if (job >= nstrips * nseg * nframes) return;
  const int strip = job % nstrips, seg = (job / nstrips) % nseg, f = job / (nstrips * nseg);
  const int t0 = seg * seg_tiles, t1 = min(t0 + seg_tiles, H / 4);
  const int x0 = strip * stride + off + 4 * lane;
  const int xl = min(max(x0, 0), W - 4);
  const uint8_t* gf = g + (size_t)f * framestride;
  u32 buf[DEPTH + 1][4];
  u32 accx = 0;
  auto ld = [&](int r) -> u32 { int rr = min(max(r, 0), H - 1); return *reinterpret_cast<const u32*>(gf + (size_t)rr * rowstride + xl); };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int k = 0; k < 4; ++k) buf[d][k] = ld(4 * (t0 + d) + k);
  for (int tb = t0; tb < t1; tb += DEPTH + 1) {
#pragma unroll
    for (int u = 0; u <= DEPTH; ++u) {
      const int t = tb + u;
      if (t < t1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) buf[(u + DEPTH) % (DEPTH + 1)][k] = ld(4 * (t + DEPTH) + k);
        accx += buf[u][0] ^ buf[u][1] ^ buf[u][2] ^ buf[u][3];
      }
    }
  }
  if (accx == 0x12345u) o[0] = 1;
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
  const size_t n = (size_t)NF * 2048 * H;   // room for padded rows
  uint8_t *s, *d; CK(hipMalloc(&s, n)); CK(hipMalloc(&d, 4096));
  CK(hipMemset(s, 7, n));
  const double bytes = (double)NF * W * H;
  float ms = timeit([&] { hipLaunchKernelGGL(rd_stream, dim3(256 * 16), dim3(256), 0, 0, (const u32x4*)s, (u32*)d, (size_t)NF * W * H / 16); });
  printf("stream read 16B/lane: %.3f ms %.0f GB/s\n", ms, bytes / ms * 1e-6);
  struct Cfg { int stride, off; } cfgs[] = { {244, -8}, {256, 0}, {256, -8}, {256, -64}, {240, -8}, {240, 0}, {192, -8}, {192, 0}, {128, 0}, {128, -64} };
  for (int nseg : { 2, 10 }) {
    const int seg_tiles = (270 + nseg - 1) / nseg;
    for (auto c : cfgs) {
      const int nstrips = (W - c.off + c.stride - 1) / c.stride;
      const int jobs = nstrips * nseg * NF;
      float m4 = timeit([&] { hipLaunchKernelGGL((rd_march<2, 4>), dim3((jobs + 3) / 4), dim3(256), 0, 0, s, d, c.stride, c.off, nstrips, nseg, seg_tiles, NF, W, W * H); });
      float m1 = timeit([&] { hipLaunchKernelGGL((rd_march<2, 1>), dim3(jobs), dim3(64), 0, 0, s, d, c.stride, c.off, nstrips, nseg, seg_tiles, NF, W, W * H); });
      float m4p = timeit([&] { hipLaunchKernelGGL((rd_march<2, 4>), dim3((jobs + 3) / 4), dim3(256), 0, 0, s, d, c.stride, c.off, nstrips, nseg, seg_tiles, NF, 2048, 2048 * H); });
      printf("nseg %2d stride %3d off %3d strips %2d : block4 %.3f ms (%.0f GB/s useful)  block1 %.3f ms   pitch2048 %.3f ms\n", nseg, c.stride, c.off, nstrips, m4, bytes / m4 * 1e-6, m1, m4p);
      fflush(stdout);
    }
  }
  return 0;
}
