// membench5.hip -- the ingest pass's ACCESS PATTERN with no arithmetic: a workgroup owns a destination tile TW x 8 and,
// frame after frame (32 per block), reads the tile's source box as row pieces of (TW + 16) * 3 bytes at the frame's row
// pitch and writes its TW x 8 grey bytes.  Does the piece length matter?  Build: hipcc --offload-arch=gfx950 -O3 -o membench5 membench5.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define W 1920
#define H 1080
#define NF 1024
#define FPB 32
// NT threads, 4 px per output quad.  Box: rows y0-1 .. y0+TH (TH + 2 rows), columns x0-8 .. x0+TW+8.
template <int TW, int TH, int NT>
__global__ __launch_bounds__(1024) void k(const uint8_t* __restrict__ in, uint8_t* __restrict__ out)
{
  constexpr int ntx = W / TW, nty = H / TH;
  const int tile = blockIdx.x % (ntx * nty), bz = blockIdx.x / (ntx * nty);
  const int bx = tile % ntx, by = tile / ntx;
  const int tid = threadIdx.x;
  const int x0 = bx * TW, y0 = by * TH;
  constexpr int upr = (TW + 16) / 4;                 // 12-byte units per box row
  constexpr int nunits = upr * (TH + 2);
  constexpr int SL = (nunits + NT - 1) / NT;
  unsigned acc = 0;
  for (int f = bz * FPB; f < bz * FPB + FPB; ++f) {
    const uint8_t* src = in + (size_t)f * W * H * 3;
#pragma unroll
    for (int s = 0; s < SL; ++s) {
      const int u = tid + NT * s;
      if (u < nunits) {
        const int r = u / upr, c = u - r * upr;
        int sx = x0 - 8 + 4 * c, sy = y0 - 1 + r;
        sx = min(max(sx, 0), W - 4); sy = min(max(sy, 0), H - 1);
        const unsigned* p = reinterpret_cast<const unsigned*>(src + ((size_t)sy * W + sx) * 3);
        acc += p[0] ^ p[1] ^ p[2];
      }
    }
    for (int q = tid; q < (TW / 4) * TH; q += NT) {
      const int tx = q % (TW / 4), ty = q / (TW / 4);
      *reinterpret_cast<unsigned*>(out + (size_t)f * W * H + (size_t)(y0 + ty) * W + x0 + 4 * tx) = acc;
    }
  }
}
int main()
{
  uint8_t *in, *out;
  if (hipMalloc(&in, (size_t)W * H * 3 * NF) != hipSuccess || hipMalloc(&out, (size_t)W * H * NF) != hipSuccess) return 1;
  (void)hipMemset(in, 1, (size_t)W * H * 3 * NF);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto run = [&](const char* name, auto kern, int tw, int th, int nt) {
    const unsigned blocks = (unsigned)((W / tw) * (H / th) * (NF / FPB));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(nt), 0, 0, in, out);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(nt), 0, 0, in, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 4;
    printf("%-34s %.3f ms  %.0f GB/s (4 bytes per pixel)\n", name, ms, 4.0 * W * H * NF / ms / 1e6); fflush(stdout);
  };
  run("tile 128 x 8, 256 threads", k<128, 8, 256>, 128, 8, 256);
  run("tile 384 x 8, 768 threads", k<384, 8, 768>, 384, 8, 768);
  run("tile 384 x 8, 256 threads", k<384, 8, 256>, 384, 8, 256);
  run("tile 640 x 8, 512 threads", k<640, 8, 512>, 640, 8, 512);
  run("tile 1920 x 8, 1024 threads", k<1920, 8, 1024>, 1920, 8, 1024);
  run("tile 128 x 24, 256 threads", k<128, 24, 256>, 128, 24, 256);
  run("tile 384 x 24, 768 threads", k<384, 24, 768>, 384, 24, 768);
  run("tile 128 x 8, 256 threads", k<128, 8, 256>, 128, 8, 256);
  return 0;
}
