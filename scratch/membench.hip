// membench.hip -- experiment: what do candidate access patterns of the threshold+corner pass cost on
// their own (loads + stores, no arithmetic)?  1024 frames 1920x1080 u8 in -> u8 out.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define W 1920
#define H 1080
typedef unsigned u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void p0_copy16(const u32x4* __restrict__ s, u32x4* __restrict__ d, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = s[i];
}
__global__ __launch_bounds__(256) void p0_copy4(const u32* __restrict__ s, u32* __restrict__ d, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = s[i];
}

// P1: wave marches a 256-B strip (244 useful), DEPTH tile rows of dword loads in flight
template <int DEPTH, int STRIDE, int MODE = 0, int LSTRIDE = STRIDE>
__global__ __launch_bounds__(256) void p1_march4(const uint8_t* __restrict__ g, uint8_t* __restrict__ o, int nseg, int seg_tiles, int nframes)
{
  const int lane = threadIdx.x & 63;
  const int job = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int nstrips = (W + STRIDE - 1) / STRIDE;
  if (job >= nstrips * nseg * nframes) return;
  const int strip = job % nstrips, seg = (job / nstrips) % nseg, f = job / (nstrips * nseg);
  const int t0 = seg * seg_tiles, t1 = min(t0 + seg_tiles, H / 4);
  const int x0 = STRIDE == 256 ? strip * 256 + 4 * lane : strip * STRIDE - 8 + 4 * lane;
  const int xl = LSTRIDE == STRIDE ? min(max(x0, 0), W - 4) : min(max(strip * LSTRIDE + 4 * lane, 0), W - 4);
  const bool lo = (STRIDE == 256 || (lane >= 2 && lane < 2 + STRIDE / 4)) && x0 >= 0 && x0 < W;
  const uint8_t* gf = g + (size_t)f * W * H;
  uint8_t* of = o + (size_t)f * W * H;
  u32 buf[DEPTH + 1][4];
  u32 accx = 0;
  auto ld = [&](int r) -> u32 { if (MODE == 2) return (u32)r; int rr = min(max(r, 0), H - 1); return *reinterpret_cast<const u32*>(gf + (size_t)rr * W + xl); };
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
        if (lo) {
#pragma unroll
          for (int k = 0; k < 4; ++k) *reinterpret_cast<u32*>(of + (size_t)(4 * t + k) * W + x0) = buf[u][k] ^ 0x55u;
        }
      }
    }
  }
}


template <int DEPTH, int STRIDE>
__global__ __launch_bounds__(256) void p2_tile16(const uint8_t* __restrict__ g, uint8_t* __restrict__ o, int nseg, int seg_tiles, int nframes)
{
  const int lane = threadIdx.x & 63;
  const int job = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int nstrips = 8;
  if (job >= nstrips * nseg * nframes) return;
  const int strip = job % nstrips, seg = (job / nstrips) % nseg, f = job / (nstrips * nseg);
  const int t0 = seg * seg_tiles, t1 = min(t0 + seg_tiles, H / 4);
  const int k = lane >> 4;
  const int x0 = (STRIDE == 256 ? strip * 256 : strip * STRIDE - 8) + 16 * (lane & 15);
  const int xl = min(max(x0, 0), W - 16);
  const bool lo = x0 >= 0 && x0 + 16 <= W;
  const uint8_t* gf = g + (size_t)f * W * H;
  uint8_t* of = o + (size_t)f * W * H;
  u32x4 buf[DEPTH + 1];
  auto ld = [&](int r) { int rr = min(max(r, 0), H - 1); return *reinterpret_cast<const u32x4*>(gf + (size_t)rr * W + xl); };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) buf[d] = ld(4 * (t0 + d) + k);
  for (int tb = t0; tb < t1; tb += DEPTH + 1) {
#pragma unroll
    for (int u = 0; u <= DEPTH; ++u) {
      const int t = tb + u;
      if (t < t1) {
        buf[(u + DEPTH) % (DEPTH + 1)] = ld(4 * (t + DEPTH) + k);
        if (lo) *reinterpret_cast<u32x4*>(of + (size_t)(4 * t + k) * W + x0) = buf[u] ^ 0x55u;
      }
    }
  }
}

// P3: block of 4 waves marches a 1024-B strip (960 useful => 2 blocks per frame row); wave j moves
// row 4t+j with one 16 B/lane access (1 KB contiguous)
template <int DEPTH>
__global__ __launch_bounds__(256) void p3_block16(const uint8_t* __restrict__ g, uint8_t* __restrict__ o, int nseg, int seg_tiles, int nframes)
{
  const int lane = threadIdx.x & 63, j = threadIdx.x >> 6;
  const int job = blockIdx.x;
  if (job >= 2 * nseg * nframes) return;
  const int strip = job % 2, seg = (job / 2) % nseg, f = job / (2 * nseg);
  const int t0 = seg * seg_tiles, t1 = min(t0 + seg_tiles, H / 4);
  const int x0 = strip * 960 - 32 + 16 * lane;
  const int xl = min(max(x0, 0), W - 16);
  const bool lo = lane >= 2 && lane <= 61 && x0 >= 0 && x0 < W;
  const uint8_t* gf = g + (size_t)f * W * H;
  uint8_t* of = o + (size_t)f * W * H;
  u32x4 buf[DEPTH + 1];
  auto ld = [&](int r) { int rr = min(max(r, 0), H - 1); return *reinterpret_cast<const u32x4*>(gf + (size_t)rr * W + xl); };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) buf[d] = ld(4 * (t0 + d) + j);
  for (int tb = t0; tb < t1; tb += DEPTH + 1) {
#pragma unroll
    for (int u = 0; u <= DEPTH; ++u) {
      const int t = tb + u;
      if (t < t1) {
        buf[(u + DEPTH) % (DEPTH + 1)] = ld(4 * (t + DEPTH) + j);
        if (lo) *reinterpret_cast<u32x4*>(of + (size_t)(4 * t + j) * W + x0) = buf[u] ^ 0x55u;
      }
    }
  }
}

// P4: one wave marches a 1024-B strip by itself: 4 x (16 B/lane) per tile row
template <int DEPTH>
__global__ __launch_bounds__(256) void p4_wave16(const uint8_t* __restrict__ g, uint8_t* __restrict__ o, int nseg, int seg_tiles, int nframes)
{
  const int lane = threadIdx.x & 63;
  const int job = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (job >= 2 * nseg * nframes) return;
  const int strip = job % 2, seg = (job / 2) % nseg, f = job / (2 * nseg);
  const int t0 = seg * seg_tiles, t1 = min(t0 + seg_tiles, H / 4);
  const int x0 = strip * 960 - 32 + 16 * lane;
  const int xl = min(max(x0, 0), W - 16);
  const bool lo = lane >= 2 && lane <= 61 && x0 >= 0 && x0 < W;
  const uint8_t* gf = g + (size_t)f * W * H;
  uint8_t* of = o + (size_t)f * W * H;
  u32x4 buf[DEPTH + 1][4];
  auto ld = [&](int r) { int rr = min(max(r, 0), H - 1); return *reinterpret_cast<const u32x4*>(gf + (size_t)rr * W + xl); };
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
        if (lo) {
#pragma unroll
          for (int k = 0; k < 4; ++k) *reinterpret_cast<u32x4*>(of + (size_t)(4 * t + k) * W + x0) = buf[u][k] ^ 0x55u;
        }
      }
    }
  }
}

// P5: wave marches a 512-B strip: 8 B/lane
template <int DEPTH>
__global__ __launch_bounds__(256) void p5_wave8(const uint8_t* __restrict__ g, uint8_t* __restrict__ o, int nseg, int seg_tiles, int nframes)
{
  typedef u32 u32x2 __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x & 63;
  const int job = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (job >= 4 * nseg * nframes) return;
  const int strip = job % 4, seg = (job / 4) % nseg, f = job / (4 * nseg);
  const int t0 = seg * seg_tiles, t1 = min(t0 + seg_tiles, H / 4);
  const int x0 = strip * 480 - 16 + 8 * lane;
  const int xl = min(max(x0, 0), W - 8);
  const bool lo = lane >= 2 && lane <= 61 && x0 >= 0 && x0 < W;
  const uint8_t* gf = g + (size_t)f * W * H;
  uint8_t* of = o + (size_t)f * W * H;
  u32x2 buf[DEPTH + 1][4];
  auto ld = [&](int r) { int rr = min(max(r, 0), H - 1); return *reinterpret_cast<const u32x2*>(gf + (size_t)rr * W + xl); };
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
        if (lo) {
#pragma unroll
          for (int k = 0; k < 4; ++k) *reinterpret_cast<u32x2*>(of + (size_t)(4 * t + k) * W + x0) = buf[u][k] ^ 0x55u;
        }
      }
    }
  }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
template <class F> void timeit(const char* name, F launch, double bytes)
{
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  launch(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < 10; ++i) launch();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
  CK(hipGetLastError());
  printf("%-28s %.3f ms  %.0f GB/s\n", name, ms, bytes / ms * 1e-6);
  fflush(stdout);
}
int main()
{
  const int NF = 1024;
  const size_t n = (size_t)NF * W * H;
  uint8_t *s, *d; CK(hipMalloc(&s, n)); CK(hipMalloc(&d, n));
  CK(hipMemset(s, 7, n)); CK(hipMemset(d, 0, n));
  const double bytes = 2.0 * n;
  timeit("p0 copy 16B/lane", [&] { hipLaunchKernelGGL(p0_copy16, dim3(256 * 16), dim3(256), 0, 0, (const u32x4*)s, (u32x4*)d, n / 16); }, bytes);
  timeit("p0 copy 4B/lane", [&] { hipLaunchKernelGGL(p0_copy4, dim3(256 * 16), dim3(256), 0, 0, (const u32*)s, (u32*)d, n / 4); }, bytes);
  for (int nseg : { 2, 10 }) {
    const int seg_tiles = (270 + nseg - 1) / nseg;
    printf("nseg %d\n", nseg);
    int jobs = 8 * nseg * NF;
    timeit(" p1 march 4B depth1", [&] { hipLaunchKernelGGL((p1_march4<1,244>), dim3((jobs + 3) / 4), dim3(256), 0, 0, s, d, nseg, seg_tiles, NF); }, bytes);
    timeit(" p1 march 4B depth2", [&] { hipLaunchKernelGGL((p1_march4<2,244>), dim3((jobs + 3) / 4), dim3(256), 0, 0, s, d, nseg, seg_tiles, NF); }, bytes);
    timeit(" p1 march 4B depth4", [&] { hipLaunchKernelGGL((p1_march4<4,244>), dim3((jobs + 3) / 4), dim3(256), 0, 0, s, d, nseg, seg_tiles, NF); }, bytes);
    timeit(" p1 ALIGNED 4B depth2", [&] { hipLaunchKernelGGL((p1_march4<2,256>), dim3((jobs + 3) / 4), dim3(256), 0, 0, s, d, nseg, seg_tiles, NF); }, bytes);
    timeit(" p1 stride240 4B depth2", [&] { hipLaunchKernelGGL((p1_march4<2,240>), dim3((jobs + 3) / 4), dim3(256), 0, 0, s, d, nseg, seg_tiles, NF); }, bytes);
    timeit(" p2 tile16 s244 depth2", [&] { hipLaunchKernelGGL((p2_tile16<2,244>), dim3((jobs + 3) / 4), dim3(256), 0, 0, s, d, nseg, seg_tiles, NF); }, bytes);
    timeit(" p2 tile16 ALIGNED depth2", [&] { hipLaunchKernelGGL((p2_tile16<2,256>), dim3((jobs + 3) / 4), dim3(256), 0, 0, s, d, nseg, seg_tiles, NF); }, bytes);
    for (int rep = 0; rep < 1; ++rep) {
      auto J = [&](int stride) { return ((W + stride - 1) / stride) * nseg * NF; };
      timeit(" loads only  s244", [&] { hipLaunchKernelGGL((p1_march4<2,244,1>), dim3((J(244) + 3) / 4), dim3(256), 0, 0, s, d, nseg, seg_tiles, NF); }, bytes / 2);
      timeit(" loads only  s256", [&] { hipLaunchKernelGGL((p1_march4<2,256,1>), dim3((J(256) + 3) / 4), dim3(256), 0, 0, s, d, nseg, seg_tiles, NF); }, bytes / 2);
      timeit(" stores only s244", [&] { hipLaunchKernelGGL((p1_march4<2,244,2>), dim3((J(244) + 3) / 4), dim3(256), 0, 0, s, d, nseg, seg_tiles, NF); }, bytes / 2);
      timeit(" stores only s256", [&] { hipLaunchKernelGGL((p1_march4<2,256,2>), dim3((J(256) + 3) / 4), dim3(256), 0, 0, s, d, nseg, seg_tiles, NF); }, bytes / 2);
      timeit(" both s192", [&] { hipLaunchKernelGGL((p1_march4<2,192>), dim3((J(192) + 3) / 4), dim3(256), 0, 0, s, d, nseg, seg_tiles, NF); }, bytes);
      timeit(" both s224", [&] { hipLaunchKernelGGL((p1_march4<2,224>), dim3((J(224) + 3) / 4), dim3(256), 0, 0, s, d, nseg, seg_tiles, NF); }, bytes);
      timeit(" both s128", [&] { hipLaunchKernelGGL((p1_march4<2,128>), dim3((J(128) + 3) / 4), dim3(256), 0, 0, s, d, nseg, seg_tiles, NF); }, bytes);
    }
    jobs = 4 * nseg * NF;
    timeit(" p5 wave 8B depth1", [&] { hipLaunchKernelGGL(p5_wave8<1>, dim3((jobs + 3) / 4), dim3(256), 0, 0, s, d, nseg, seg_tiles, NF); }, bytes);
    timeit(" p5 wave 8B depth2", [&] { hipLaunchKernelGGL(p5_wave8<2>, dim3((jobs + 3) / 4), dim3(256), 0, 0, s, d, nseg, seg_tiles, NF); }, bytes);
    jobs = 2 * nseg * NF;
    timeit(" p3 block 16B depth1", [&] { hipLaunchKernelGGL(p3_block16<1>, dim3(jobs), dim3(256), 0, 0, s, d, nseg, seg_tiles, NF); }, bytes);
    timeit(" p3 block 16B depth2", [&] { hipLaunchKernelGGL(p3_block16<2>, dim3(jobs), dim3(256), 0, 0, s, d, nseg, seg_tiles, NF); }, bytes);
    timeit(" p3 block 16B depth4", [&] { hipLaunchKernelGGL(p3_block16<4>, dim3(jobs), dim3(256), 0, 0, s, d, nseg, seg_tiles, NF); }, bytes);
    timeit(" p4 wave 16B depth1", [&] { hipLaunchKernelGGL(p4_wave16<1>, dim3((jobs + 3) / 4), dim3(256), 0, 0, s, d, nseg, seg_tiles, NF); }, bytes);
    timeit(" p4 wave 16B depth2", [&] { hipLaunchKernelGGL(p4_wave16<2>, dim3((jobs + 3) / 4), dim3(256), 0, 0, s, d, nseg, seg_tiles, NF); }, bytes);
  }
  return 0;
}
