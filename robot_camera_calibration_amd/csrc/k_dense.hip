// k_dense.hip -- the "threshold + corner pass": stages a3 (apriltag tile threshold) and a4.1
// (integer Harris response on the even pixel lattice, 3x3 lattice maximum selection, wave-ballot
// compaction), fused: one read of the
// grey image, one write of the threshold map, a short candidate list.
//
// Stands in for the per-pixel passes of the external detector (apriltag threshold(); SURVEY.md
// section 3.1) whose output the reference consumes at corner_detections.cpp:41-56.  Definitions:
// SURVEY appendix B.3 (threshold) and DESIGN.md section 3 (integer Harris, k = 1/16).
//
// Algorithmic HBM bytes: 2 B/px (1 read + 1 written) = 4.147 MB per 1080p frame.
//
// Variant 0 (this kernel, k_dense_lds): any geometry.  One 256-thread block per 64x32 pixel
// block of 4x4 threshold tiles (the last block of a row/column absorbs the <=3 ragged pixels);
// the grey region with a 4-pixel halo is staged in LDS once and every stencil stage reads LDS.
#include "rcc_internal.h"

#define BW 64
#define BH 32
#define OFF 8              // region origin = (x0-OFF, y0-OFF): halo 5 needed, 8 keeps rows dword-aligned
#define GR_W 84            // >= BW + 3 + OFF + 7
#define GR_H (BH + 3 + OFF + 7)
#define GD_W (BW + 3 + 8)  // gradient region [x0-4, x1+4)
#define GD_H (BH + 3 + 8)
#define LT_W ((BW + 3 + 4) / 2 + 2)  // lattice columns covering [x0-2, x1+2)
#define LT_H ((BH + 3 + 4) / 2 + 2)
#define TL_W 18            // threshold tiles incl. halo
#define TL_H 10

__global__ __launch_bounds__(256) void k_dense_lds(const uint8_t* __restrict__ grey, int w, int h,
                                                   int nbx, int nby, int min_contrast, int hthresh,
                                                   int margin, int cap, uint8_t* __restrict__ bin,
                                                   rcc_cand* __restrict__ cand, int32_t* __restrict__ cand_count)
{
  __shared__ uint8_t sg[GR_H][GR_W];
  __shared__ int8_t sgx[GD_H][GD_W], sgy[GD_H][GD_W];
  __shared__ int32_t shA[GD_H][LT_W], shB[GD_H][LT_W], shC[GD_H][LT_W];   // horizontal raw 5-sums at even x
  __shared__ int32_t sR[LT_H][LT_W];
  __shared__ uint8_t stmin[TL_H][TL_W], stmax[TL_H][TL_W];
  __shared__ uint8_t sdmin[TL_H][TL_W], sdmax[TL_H][TL_W];

  const int tid = threadIdx.x;
  const int f = blockIdx.z;
  const int bx = blockIdx.x, by = blockIdx.y;
  int tw = w >> 2, th = h >> 2;
  if (tw < 1) tw = 1;
  if (th < 1) th = 1;
  const int x0 = bx * BW, y0 = by * BH;
  const int x1 = (bx == nbx - 1) ? w : x0 + BW;   // exclusive
  const int y1 = (by == nby - 1) ? h : y0 + BH;
  const int bw = x1 - x0, bh = y1 - y0;           // <= 67, <= 35
  const uint8_t* g = grey + (size_t)f * w * h;
  uint8_t* bo = bin + (size_t)f * w * h;

  // ---- stage 0: grey region [x0-8, x1+7) x [y0-8, y1+7), coordinates clamped to the image
  // (+7 right/bottom: 5 for the lattice halo, and an extended last tile seen as a halo tile)
  const int rw = bw + OFF + 7, rh = bh + OFF + 7;
  for (int i = tid; i < rw * rh; i += 256) {
    int ry = i / rw, rx = i - ry * rw;
    int gx = min(max(x0 - OFF + rx, 0), w - 1), gy = min(max(y0 - OFF + ry, 0), h - 1);
    sg[ry][rx] = g[(size_t)gy * w + gx];
  }
  __syncthreads();

  // ---- stage 1: threshold-tile min/max for tiles [tx0-1, ...] (halo of one tile), then 3x3 dilation
  const int tx0 = x0 >> 2, ty0 = y0 >> 2;
  const int ntx = ((bx == nbx - 1) ? tw - tx0 : BW / 4), nty = ((by == nby - 1) ? th - ty0 : BH / 4);
  for (int i = tid; i < TL_W * TL_H; i += 256) {
    int j = i / TL_W, k = i - j * TL_W;
    int tx = tx0 - 1 + k, ty = ty0 - 1 + j;
    int mn = 255, mx = 0;
    if (k < ntx + 2 && j < nty + 2 && tx >= 0 && tx < tw && ty >= 0 && ty < th) {
      int px0 = tx * 4, px1 = (tx == tw - 1) ? w : px0 + 4;
      int py0 = ty * 4, py1 = (ty == th - 1) ? h : py0 + 4;
      for (int yy = py0; yy < py1; ++yy)
        for (int xx = px0; xx < px1; ++xx) {
          int v = sg[yy - (y0 - OFF)][xx - (x0 - OFF)];
          mn = min(mn, v);
          mx = max(mx, v);
        }
    }
    stmin[j][k] = (uint8_t)mn;
    stmax[j][k] = (uint8_t)mx;
  }
  __syncthreads();
  for (int i = tid; i < TL_W * TL_H; i += 256) {
    int j = i / TL_W, k = i - j * TL_W;
    int mn = 255, mx = 0;
    if (j >= 1 && j <= nty && k >= 1 && k <= ntx) {
      for (int dj = -1; dj <= 1; ++dj)
        for (int dk = -1; dk <= 1; ++dk) {
          mn = min(mn, (int)stmin[j + dj][k + dk]);
          mx = max(mx, (int)stmax[j + dj][k + dk]);
        }
    }
    sdmin[j][k] = (uint8_t)mn;
    sdmax[j][k] = (uint8_t)mx;
  }

  // ---- stage 2: gradients on [x0-4, x1+4) x [y0-4, y1+4)
  const int dw = bw + 8, dh = bh + 8;
  for (int i = tid; i < dw * dh; i += 256) {
    int py = i / dw, px = i - py * dw;
    const int ry = py + OFF - 4, rx = px + OFF - 4;   // region coordinates of this pixel
    int a00 = sg[ry - 1][rx - 1], a01 = sg[ry - 1][rx], a02 = sg[ry - 1][rx + 1];
    int a10 = sg[ry][rx - 1], a12 = sg[ry][rx + 1];
    int a20 = sg[ry + 1][rx - 1], a21 = sg[ry + 1][rx], a22 = sg[ry + 1][rx + 1];
    int sx = (a02 + 2 * a12 + a22) - (a00 + 2 * a10 + a20);
    int sy = (a20 + 2 * a21 + a22) - (a00 + 2 * a01 + a02);
    sgx[py][px] = (int8_t)(sx >> 3);
    sgy[py][px] = (int8_t)(sy >> 3);
  }
  __syncthreads();

  // ---- stage 3: threshold map for the block's pixels
  for (int i = tid; i < bw * bh; i += 256) {
    int py = i / bw, px = i - py * bw;
    int x = x0 + px, y = y0 + py;
    int k = min(x >> 2, tw - 1) - tx0 + 1, j = min(y >> 2, th - 1) - ty0 + 1;
    int mn = sdmin[j][k], mx = sdmax[j][k];
    int v = sg[py + OFF][px + OFF];
    uint8_t o = (mx - mn < min_contrast) ? (uint8_t)127 : ((v > mn + (mx - mn) / 2) ? (uint8_t)255 : (uint8_t)0);
    bo[(size_t)y * w + x] = o;
  }

  // ---- stage 4: horizontal raw 5-sums of products at even x in [x0-2, x1+2), all gradient rows
  // lattice column c <-> x = x0 - 2 + 2c   (x0 is a multiple of 64, so these are even)
  const int lw = (bw + 4 + 1) / 2, lh = (bh + 4 + 1) / 2;
  for (int i = tid; i < lw * dh; i += 256) {
    int py = i / lw, c = i - py * lw;
    int gxc = 2 * c + 2;   // gradient-region column of x = x0-2+2c  (region starts at x0-4)
    int a = 0, b = 0, cc = 0;
#pragma unroll
    for (int d = -2; d <= 2; ++d) {
      int u = sgx[py][gxc + d], v = sgy[py][gxc + d];
      a += u * u; b += u * v; cc += v * v;
    }
    shA[py][c] = a; shB[py][c] = b; shC[py][c] = cc;
  }
  __syncthreads();

  // ---- stage 5: vertical 5-sums at even y in [y0-2, y1+2) -> response on the lattice
  for (int i = tid; i < lw * lh; i += 256) {
    int r = i / lw, c = i - r * lw;
    int x = x0 - 2 + 2 * c, y = y0 - 2 + 2 * r;
    int32_t resp = INT32_MIN;
    if (x >= 3 && x < w - 3 && y >= 3 && y < h - 3) {
      int gyr = 2 * r + 2;   // gradient-region row of y
      int sa = 0, sb = 0, sc = 0;
#pragma unroll
      for (int d = -2; d <= 2; ++d) { sa += shA[gyr + d][c]; sb += shB[gyr + d][c]; sc += shC[gyr + d][c]; }
      int A = sa >> 4, B = sb >> 4, C = sc >> 4;
      uint32_t tr = (uint32_t)(A + C);
      resp = A * C - B * B - (int32_t)((tr * tr) >> 4);
    }
    sR[r][c] = resp;
  }
  __syncthreads();

  // ---- stage 6: 3x3 maximum selection on the lattice + wave-ballot compaction
  if (margin < 6) margin = 6;
  const int cw = (bw + 1) / 2, ch = (bh + 1) / 2;   // lattice points inside the block
  const int npx = cw * ch;
  for (int base = 0; base < npx; base += 256) {
    int i = base + tid;
    bool is = false;
    int x = 0, y = 0;
    int32_t r = 0;
    if (i < npx) {
      int pr = i / cw, pc = i - pr * cw;
      x = x0 + 2 * pc; y = y0 + 2 * pr;
      const int rr = pr + 1, rc = pc + 1;
      r = sR[rr][rc];
      if (r >= hthresh && x >= margin && x < w - margin && y >= margin && y < h - margin) {
        is = r > sR[rr - 1][rc - 1] && r > sR[rr - 1][rc] && r > sR[rr - 1][rc + 1] && r > sR[rr][rc - 1] &&
             r >= sR[rr][rc + 1] && r >= sR[rr + 1][rc - 1] && r >= sR[rr + 1][rc] && r >= sR[rr + 1][rc + 1];
      }
    }
    unsigned long long m = __ballot(is);
    if (m) {
      const int lane = tid & 63;
      int basei = 0;
      if (lane == 0) basei = atomicAdd(&cand_count[f], __popcll(m));
      basei = __shfl(basei, 0);
      if (is) {
        int idx = basei + __popcll(m & ((1ull << lane) - 1ull));
        if (idx < cap) {
          rcc_cand e;
          e.x = (int16_t)x; e.y = (int16_t)y; e.score = r;
          cand[(size_t)f * cap + idx] = e;
        }
      }
    }
  }
}

hipError_t rcc_launch_dense_march(rcc_handle* h, const uint8_t* d_grey, int nframes, uint8_t* d_bin,
                                  rcc_cand* d_cand, int32_t* d_cand_count, hipStream_t s);
bool rcc_dense_march_supported(const rcc_handle* h);
hipError_t rcc_launch_dense_band(rcc_handle* h, const uint8_t* d_grey, int nframes, uint8_t* d_bin,
                                 rcc_cand* d_cand, int32_t* d_cand_count, hipStream_t s, int split);
bool rcc_dense_band_supported(const rcc_handle* h, const uint8_t* d_grey, const uint8_t* d_bin);

// variants: 0 = generic LDS tiles (any geometry), 1 = band kernel (k_dense_band.hip), 2 = strip march
// (k_dense_fast.hip), 3 = band sweep + corner kernel on the active rows (k_dense_band.hip SPLIT + k_dense_runs.hip),
// 4 = one independent wavefront per window for the compact-map form (k_dense_wave.hip; the band kernel otherwise);
// -1 = the fastest one the geometry allows.  All produce identical outputs.
hipError_t rcc_launch_dense(rcc_handle* h, const uint8_t* d_grey, int nframes, uint8_t* d_bin,
                            rcc_cand* d_cand, int32_t* d_cand_count, hipStream_t s)
{
  const rcc_config& c = h->cfg;
  if (nframes <= 0) return hipSuccess;
  hipError_t e = hipMemsetAsync(d_cand_count, 0, sizeof(int32_t) * (size_t)nframes, s);
  if (e != hipSuccess) return e;
  int variant = h->dense_variant;
#ifndef RCC_EXPERIMENTS
  if (variant == 3) variant = 1;      // the two-kernel form lives in librcc_hip_exp.so only (include/rcc_debug.h); same results
#endif
  const bool band_ok = rcc_dense_band_supported(h, d_grey, d_bin), march_ok = rcc_dense_march_supported(h);
  if (variant < 0) variant = band_ok ? 4 : (march_ok ? 2 : 0);     // 4: a wavefront per window where the compact map is asked for, else the band kernel
  if ((variant == 1 || variant == 3 || variant == 4) && !band_ok) variant = march_ok ? 2 : 0;
  if (variant == 2 && !march_ok) variant = 0;
  h->bin_from_thr = 0;      // only the band kernel can leave the binary image as the compact threshold map
  if (variant == 1 || variant == 3 || variant == 4) return rcc_launch_dense_band(h, d_grey, nframes, d_bin, d_cand, d_cand_count, s, variant == 3 ? 1 : variant == 4 ? 2 : 0);
  if (variant == 2) { h->dense_kernel = "k_dense_march<0>"; return rcc_launch_dense_march(h, d_grey, nframes, d_bin, d_cand, d_cand_count, s); }
  h->dense_kernel = "k_dense_lds";
  const int w = c.width, ht = c.height;
  int tw = w >> 2, th = ht >> 2;
  if (tw < 1) tw = 1;
  if (th < 1) th = 1;
  const int nbx = (tw * 4 + BW - 1) / BW, nby = (th * 4 + BH - 1) / BH;
  dim3 grid(nbx, nby, nframes);
  hipLaunchKernelGGL(k_dense_lds, grid, dim3(256), 0, s, d_grey, w, ht, nbx, nby, c.thr_min_contrast,
                     c.harris_thresh, c.cand_margin, c.max_candidates, d_bin, d_cand, d_cand_count);
  return hipGetLastError();
}
