// k_ingest.hip -- stage a1+a2: BGR8/MONO8 -> grey u8, optional per-pixel undistortion.
//
// Stands in for the image callback of the external detector node the reference launches on
// image_raw (real_preprocessing/README.md:65; cv_bridge dependency real_preprocessing/package.xml:56).
// Arithmetic: SURVEY.md appendix B.1 (grey) and B.2 (initUndistortRectifyMap + remap, Q5 map,
// bilinear, constant border); intrinsics laid out as camera_pose.cpp:59-64 loads them.
// Order is grey first, then remap of the grey image.
//
// HBM view (per 1080p BGR frame): 6.22 MB read + 2.07 MB written = 4 B/px algorithmic.
// The map is never stored: it is recomputed per destination pixel in fp64 and amortised over the
// frames a block walks (frames_per_block), so it costs no traffic.
//
// Variants:
//   0  gather: taps read straight from global memory (any geometry)
//   1  staged: the source rows a destination tile needs are loaded coalesced (16 B/lane),
//      converted to grey once, kept in LDS, taps come from LDS
#include "ingest_staged.h"

// ---- variant 0: gather ------------------------------------------------------------------------
// block (64,4): 64 quads of 4 pixels x 4 rows; grid (ceil(w/256), ceil(h/4), ceil(nframes/fpb))
template <int NCH, bool RGB = false>
__global__ __launch_bounds__(256) void k_ingest_gather(const uint8_t* __restrict__ frames,
                                                       int64_t frame_bytes, int stride, int w, int h,
                                                       rcc_cam cam, uint8_t* __restrict__ grey,
                                                       int nframes, int fpb)
{
  const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
  const int y = blockIdx.y * 4 + threadIdx.y;
  if (y >= h || x0 >= w) return;
  int32_t X[4], Y[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    X[j] = 0; Y[j] = 0;
    if (x0 + j < w) rcc_map_q5(cam, x0 + j, y, X[j], Y[j]);
  }
  const int f0 = blockIdx.z * fpb;
  const int f1 = min(f0 + fpb, nframes);
  const bool vec = ((w & 3) == 0);
  for (int f = f0; f < f1; ++f) {
    const uint8_t* src = frames + (size_t)f * frame_bytes;
    uint8_t* dst = grey + (size_t)f * w * h + (size_t)y * w + x0;
    uint32_t out = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (x0 + j < w) {
        int ix = X[j] >> 5, iy = Y[j] >> 5, fx = X[j] & 31, fy = Y[j] & 31;
        int p00 = rcc_tap<NCH, RGB>(src, stride, w, h, ix, iy);
        int p01 = rcc_tap<NCH, RGB>(src, stride, w, h, ix + 1, iy);
        int p10 = rcc_tap<NCH, RGB>(src, stride, w, h, ix, iy + 1);
        int p11 = rcc_tap<NCH, RGB>(src, stride, w, h, ix + 1, iy + 1);
        int acc = (32 - fx) * (32 - fy) * p00 + fx * (32 - fy) * p01 + (32 - fx) * fy * p10 + fx * fy * p11;
        out |= (uint32_t)((acc + 512) >> 10) << (8 * j);
      }
    }
    if (vec) {
      *reinterpret_cast<uint32_t*>(dst) = out;
    } else {
      for (int j = 0; j < 4 && x0 + j < w; ++j) dst[j] = (uint8_t)(out >> (8 * j));
    }
  }
}

template <int NCH, bool RGB = false, int TH = ST_TH>
__global__ __launch_bounds__(32 * TH) void k_ingest_staged(const uint8_t* __restrict__ frames,
                                                       int64_t frame_bytes, int stride, int w, int h,
                                                       rcc_cam cam, uint8_t* __restrict__ grey,
                                                       int nframes, int fpb, int ntx, int ntiles, int per_xcd,
                                                       const int2* __restrict__ map, const int4* __restrict__ tilebox)
{
  // XCD-aware tile order.  Workgroups go round-robin to the 8 XCDs (id % 8), each with its own L2; the source
  // boxes of neighbouring tiles overlap (about 14 source rows for 8 destination rows), so neighbours must share
  // an L2 or the overlap is fetched again from the fabric (measured: 4.3x the algorithmic read bytes with the
  // plain (x, y, z) grid).  XCD c walks, for each frame group, the contiguous raster range
  // [c * per_xcd, (c + 1) * per_xcd) of tiles.
  const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
  const int bz = kk / per_xcd;
  const int tile = xcd * per_xcd + (kk - bz * per_xcd);
  if (tile >= ntiles) return;                       // block-uniform
  __shared__ __attribute__((aligned(16))) uint8_t lds[ST_TILE_LDS_T(TH)];
  __shared__ int s_flag[2];
  ingest_staged_body<NCH, RGB, TH>(frames, frame_bytes, stride, w, h, cam, grey, nframes, fpb, ntx, tile, bz, threadIdx.x, lds, s_flag, 0, 1, map, tilebox);
}

// The frame-invariant part of the staged pass, once per handle: map[v * w + u] = Q5 source coordinates of destination
// pixel (u, v) (rcc_map_q5, the arithmetic of the specification), and per 128 x 8 destination tile the bounding box of
// the integer source coordinates (min x, max x, min y, max y) -- what ingest_staged_body otherwise recomputes per block.
template <int TH>
__global__ __launch_bounds__(32 * TH) void k_ingest_map(int w, int h, rcc_cam cam, int ntx, int2* __restrict__ map, int4* __restrict__ tilebox)
{
  __shared__ int s_red[TH / 2][4];
  const int tile = blockIdx.x, tid = threadIdx.x;
  const int by = tile / ntx, bx = tile - by * ntx;
  const int tx = tid & 31, ty = tid >> 5;
  const int x0 = bx * ST_TW + tx * 4, y = by * TH + ty;
  const bool inside = (y < h) && (x0 < w);
  int mnx = INT32_MAX, mxx = INT32_MIN, mny = INT32_MAX, mxy = INT32_MIN;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (inside) {
      int32_t X, Y;
      rcc_map_q5(cam, x0 + j, y, X, Y);
      map[(size_t)y * w + x0 + j] = make_int2(X, Y);
      mnx = min(mnx, X >> 5); mxx = max(mxx, X >> 5);
      mny = min(mny, Y >> 5); mxy = max(mxy, Y >> 5);
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    mnx = min(mnx, __shfl_xor(mnx, off, 64)); mxx = max(mxx, __shfl_xor(mxx, off, 64));
    mny = min(mny, __shfl_xor(mny, off, 64)); mxy = max(mxy, __shfl_xor(mxy, off, 64));
  }
  if ((tid & 63) == 0) { s_red[tid >> 6][0] = mnx; s_red[tid >> 6][1] = mxx; s_red[tid >> 6][2] = mny; s_red[tid >> 6][3] = mxy; }
  __syncthreads();
  if (tid == 0) {
    int4 b = make_int4(s_red[0][0], s_red[0][1], s_red[0][2], s_red[0][3]);
    for (int q = 1; q < TH / 2; ++q) { b.x = min(b.x, s_red[q][0]); b.y = max(b.y, s_red[q][1]); b.z = min(b.z, s_red[q][2]); b.w = max(b.w, s_red[q][3]); }
    tilebox[tile] = b;
  }
}

// ---- no undistortion: pure streaming conversion -----------------------------------------------
// one thread per 16-pixel chunk: 3 x 16 B loads -> 1 x 16 B store
template <bool RGB>
__global__ __launch_bounds__(256) void k_grey_bgr_stream(const uint8_t* __restrict__ frames,
                                                         int64_t frame_bytes, int stride, int w, int h,
                                                         uint8_t* __restrict__ grey, int nframes, int aligned)
{
  const int cpr = (w + 15) >> 4;
  const int64_t total = (int64_t)nframes * h * cpr;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i % cpr);
    int64_t t = i / cpr;
    int y = (int)(t % h);
    int f = (int)(t / h);
    const uint8_t* src = frames + (size_t)f * frame_bytes + (size_t)y * stride + (size_t)c * 48;
    uint8_t* dst = grey + (size_t)f * w * h + (size_t)y * w + (size_t)c * 16;
    if (aligned && c * 16 + 16 <= w) {
      const uint4* s4 = reinterpret_cast<const uint4*>(src);
      uint4 a = s4[0], b = s4[1], d = s4[2];
      *reinterpret_cast<uint4*>(dst) = rcc_grey16<RGB>(a, b, d);
    } else {
      int n = min(16, w - c * 16);
      for (int j = 0; j < n; ++j) dst[j] = (uint8_t)rcc_grey_of(src[3 * j + (RGB ? 2 : 0)], src[3 * j + 1], src[3 * j + (RGB ? 0 : 2)]);
    }
  }
}

// mono, no undistortion: a copy that drops the row padding; one 16-byte chunk per thread where everything is 16-byte
// aligned, bytes otherwise
__global__ __launch_bounds__(256) void k_copy_mono(const uint8_t* __restrict__ frames, int64_t frame_bytes,
                                                   int stride, int w, int h, uint8_t* __restrict__ grey, int nframes, int aligned)
{
  const int cpr = (w + 15) >> 4;
  const int64_t total = (int64_t)nframes * h * cpr;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i % cpr);
    int64_t t = i / cpr;
    int y = (int)(t % h);
    int f = (int)(t / h);
    const uint8_t* src = frames + (size_t)f * frame_bytes + (size_t)y * stride + (size_t)c * 16;
    uint8_t* dst = grey + (size_t)f * w * h + (size_t)y * w + (size_t)c * 16;
    if (aligned && c * 16 + 16 <= w) {
      *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
    } else {
      int n = min(16, w - c * 16);
      for (int j = 0; j < n; ++j) dst[j] = src[j];
    }
  }
}

static rcc_cam make_cam(const rcc_config& c)
{
  rcc_cam k;
  k.fx = c.K[0]; k.cx = c.K[2]; k.fy = c.K[4]; k.cy = c.K[5];
  for (int i = 0; i < 8; ++i) k.D[i] = c.D[i];
  k.model = c.dist_model;
  k.solver = 0;
  return k;
}

hipError_t rcc_ingest_staged_plan(rcc_handle* h, const uint8_t* d_frames, int nframes, hipStream_t s, rcc_ingest_plan* p, bool* staged)
{
  const rcc_config& c = h->cfg;
  const int w = c.width, ht = c.height;
  *staged = false;
  if (!h->undist || nframes <= 0) return hipSuccess;
  int variant = h->ingest_variant;
  const bool staged_ok = ((w % ST_TW) == 0) && ((ht % ST_TH) == 0) && ht >= 16 && ((c.stride_bytes & 15) == 0) && ((c.frame_bytes & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(d_frames) & 15) == 0);
  if (variant < 0) variant = staged_ok ? 1 : 0;
  if (variant != 1 || !staged_ok) return hipSuccess;
  p->cam = make_cam(c);
  // tile height: 16 rows (1.16 instead of 1.25 source rows per destination row; measured on 256 x 4K fisheye: 1.735 -> 1.589 ms),
  // unless the 8-row form is asked for (rcc_set_ingest_variant(3): A/B, tests)
#ifdef RCC_EXPERIMENTS
  // the experiments library also carries the 128 x 8 form (rcc_set_ingest_variant(3), RCC_INGEST_TH=8: rounds 1-3, the A/B partner)
  // and a 128 x 32 form (RCC_INGEST_TH=32: measured, not kept)
  static const int th_env = getenv("RCC_INGEST_TH") ? atoi(getenv("RCC_INGEST_TH")) : 0;
  const int TH = (h->ingest_tile8 || th_env == 8) ? ST_TH : (th_env == 32 && ht >= 32) ? 32 : 16;
#else
  const int TH = 16;
#endif       // (a height of 8 mod 16: the last tile row's lower half is idle)
  p->th = TH;
#ifdef RCC_EXPERIMENTS
  static const int fpb_max = getenv("RCC_INGEST_FPB") ? atoi(getenv("RCC_INGEST_FPB")) : 32;
#else
  const int fpb_max = 32;
#endif
  int fpb = fpb_max;
  const int tiles = ((w + ST_TW - 1) / ST_TW) * ((ht + TH - 1) / TH);
  while (fpb > 2 && (int64_t)tiles * ((nframes + fpb - 1) / fpb) < 4096 * ST_TH / TH) fpb >>= 1;
  p->fpb = fpb; p->tiles = tiles;
  p->ntx = (w + ST_TW - 1) / ST_TW; p->per_xcd = (tiles + 7) / 8; p->ngroups = (nframes + fpb - 1) / fpb;
  if ((!h->d_map || h->map_th != TH) && !h->map_failed) {
    // first staged launch of this handle (or the first with this tile height): tabulate the map (w * h * 8 B) and the tiles'
    // source boxes
    if (h->d_tilebox) { (void)hipStreamSynchronize(s); (void)hipFree(h->d_tilebox); h->d_tilebox = nullptr; }
    if ((!h->d_map && hipMalloc((void**)&h->d_map, (size_t)w * ht * sizeof(int2)) != hipSuccess) || hipMalloc((void**)&h->d_tilebox, (size_t)tiles * sizeof(int4)) != hipSuccess) {
      if (h->d_map) (void)hipFree(h->d_map);
      h->d_map = nullptr; h->d_tilebox = nullptr; h->map_failed = 1;      // no room: the kernel recomputes the map itself
      (void)hipGetLastError();
    } else {
#ifdef RCC_EXPERIMENTS
      if (TH == 32) hipLaunchKernelGGL(k_ingest_map<32>, dim3(tiles), dim3(1024), 0, s, w, ht, p->cam, p->ntx, (int2*)h->d_map, (int4*)h->d_tilebox);
      else if (TH == ST_TH) hipLaunchKernelGGL(k_ingest_map<ST_TH>, dim3(tiles), dim3(256), 0, s, w, ht, p->cam, p->ntx, (int2*)h->d_map, (int4*)h->d_tilebox);
      else
#endif
      hipLaunchKernelGGL(k_ingest_map<16>, dim3(tiles), dim3(512), 0, s, w, ht, p->cam, p->ntx, (int2*)h->d_map, (int4*)h->d_tilebox);
      hipError_t em = hipGetLastError();
      if (em == hipSuccess) em = hipStreamSynchronize(s);      // once per handle: later launches may come on other streams
      if (em != hipSuccess) {
        // tables that may not have been written must never be read: drop them, the kernel recomputes the map per block
        (void)hipFree(h->d_map); (void)hipFree(h->d_tilebox);
        h->d_map = nullptr; h->d_tilebox = nullptr; h->map_failed = 1;
        return em;
      }
      h->map_th = TH;
    }
  }
  p->map = h->ingest_table ? h->d_map : nullptr;
  p->tilebox = h->ingest_table ? h->d_tilebox : nullptr;
  if (!p->map || !p->tilebox) { p->map = nullptr; p->tilebox = nullptr; }
  *staged = true;
  return hipSuccess;
}

hipError_t rcc_launch_ingest(rcc_handle* h, const uint8_t* d_frames, int nframes, uint8_t* d_grey, hipStream_t s)
{
  const rcc_config& c = h->cfg;
  const int w = c.width, ht = c.height;
  if (nframes <= 0) return hipSuccess;
  if (!h->undist) {
    if (c.pixfmt != RCC_PIX_MONO8) {
      int aligned = ((c.stride_bytes & 15) == 0) && ((c.frame_bytes & 15) == 0) && ((w & 15) == 0) &&
                    ((reinterpret_cast<uintptr_t>(d_frames) & 15) == 0) && ((reinterpret_cast<uintptr_t>(d_grey) & 15) == 0);
      int64_t total = (int64_t)nframes * ht * ((w + 15) >> 4);
      // one 16-pixel chunk per thread (the grid-stride loop only serves batches beyond 2^31 chunks): as many independent
      // loads in flight as the device holds waves: 1.72 -> 1.44 ms per 1024 x 1080p, the rate of the bare byte movement
      // (scratch/membench4.hip); it ran as 4096 blocks of a grid-stride loop before
      int blocks = (int)(((total + 255) / 256) < 0x7FFFFFFF ? ((total + 255) / 256) : 0x7FFFFFFF);
      if (c.pixfmt == RCC_PIX_RGB8)
        hipLaunchKernelGGL(k_grey_bgr_stream<true>, dim3(blocks), dim3(256), 0, s, d_frames, c.frame_bytes, c.stride_bytes, w, ht, d_grey, nframes, aligned);
      else
        hipLaunchKernelGGL(k_grey_bgr_stream<false>, dim3(blocks), dim3(256), 0, s, d_frames, c.frame_bytes, c.stride_bytes, w, ht, d_grey, nframes, aligned);
    } else {
      int aligned = ((c.stride_bytes & 15) == 0) && ((c.frame_bytes & 15) == 0) && ((w & 15) == 0) &&
                    ((reinterpret_cast<uintptr_t>(d_frames) & 15) == 0) && ((reinterpret_cast<uintptr_t>(d_grey) & 15) == 0);
      int64_t total = (int64_t)nframes * ht * ((w + 15) >> 4);
      int blocks = (int)(((total + 255) / 256) < 0x7FFFFFFF ? ((total + 255) / 256) : 0x7FFFFFFF);
      hipLaunchKernelGGL(k_copy_mono, dim3(blocks), dim3(256), 0, s, d_frames, c.frame_bytes, c.stride_bytes, w, ht, d_grey, nframes, aligned);
    }
    return hipGetLastError();
  }
  {
    rcc_ingest_plan p;
    bool staged = false;
    hipError_t e = rcc_ingest_staged_plan(h, d_frames, nframes, s, &p, &staged);
    if (e != hipSuccess) return e;
    if (staged) {
      dim3 grid(8 * p.per_xcd * p.ngroups);
#define LAUNCH_STAGED(NCH_, RGB_, TH_) hipLaunchKernelGGL((k_ingest_staged<NCH_, RGB_, TH_>), grid, dim3(32 * TH_), 0, s, d_frames, c.frame_bytes, c.stride_bytes, w, ht, \
                                                          p.cam, d_grey, nframes, p.fpb, p.ntx, p.tiles, p.per_xcd, (const int2*)p.map, (const int4*)p.tilebox)
#ifdef RCC_EXPERIMENTS
      if (p.th == 32) {
        if (c.pixfmt == RCC_PIX_RGB8) LAUNCH_STAGED(3, true, 32);
        else if (c.pixfmt == RCC_PIX_BGR8) LAUNCH_STAGED(3, false, 32);
        else LAUNCH_STAGED(1, false, 32);
      } else if (p.th == ST_TH) {
        if (c.pixfmt == RCC_PIX_RGB8) LAUNCH_STAGED(3, true, ST_TH);
        else if (c.pixfmt == RCC_PIX_BGR8) LAUNCH_STAGED(3, false, ST_TH);
        else LAUNCH_STAGED(1, false, ST_TH);
      } else
#endif
      {
        if (c.pixfmt == RCC_PIX_RGB8) LAUNCH_STAGED(3, true, 16);
        else if (c.pixfmt == RCC_PIX_BGR8) LAUNCH_STAGED(3, false, 16);
        else LAUNCH_STAGED(1, false, 16);
      }
#undef LAUNCH_STAGED
      return hipGetLastError();
    }
  }
  rcc_cam cam = make_cam(c);
  // frames per block: amortise the fp64 map; keep >= ~2048 blocks in flight
  int fpb = 16;
  const int tiles = ((w + 255) / 256) * ((ht + 3) / 4);
  while (fpb > 1 && (int64_t)tiles * ((nframes + fpb - 1) / fpb) < 2048) fpb >>= 1;
  dim3 grid((w + 255) / 256, (ht + 3) / 4, (nframes + fpb - 1) / fpb), block(64, 4);
  if (c.pixfmt == RCC_PIX_RGB8)
    hipLaunchKernelGGL((k_ingest_gather<3, true>), grid, block, 0, s, d_frames, c.frame_bytes, c.stride_bytes, w, ht, cam, d_grey, nframes, fpb);
  else if (c.pixfmt == RCC_PIX_BGR8)
    hipLaunchKernelGGL((k_ingest_gather<3>), grid, block, 0, s, d_frames, c.frame_bytes, c.stride_bytes, w, ht, cam, d_grey, nframes, fpb);
  else
    hipLaunchKernelGGL((k_ingest_gather<1>), grid, block, 0, s, d_frames, c.frame_bytes, c.stride_bytes, w, ht, cam, d_grey, nframes, fpb);
  return hipGetLastError();
}
