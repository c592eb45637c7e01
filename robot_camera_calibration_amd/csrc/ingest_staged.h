// ingest_staged.h -- stage a1+a2 arithmetic and the LDS-staged tile body of the undistort+grey pass, shared by
// k_ingest.hip (the pass as its own kernel) and k_mix.hip (the pass side by side with the threshold+corner pass of the
// previous chunk in one launch).  See k_ingest.hip for the definitions and the measured history.
#pragma once
#include "rcc_internal.h"

// ---- map ------------------------------------------------------------------------------------
// Bit-for-bit the operation sequence of the specification (one rounded IEEE op per line;
// this translation unit is compiled with -ffp-contract=off).
__device__ __forceinline__ double rcc_atan_pos(double r)
{
  const double PI_2 = 1.57079632679489661923, PI_4 = 0.78539816339744830962;
  const double T8 = 0.41421356237309504880;
  bool flip = false;
  double t = r;
  if (t > 1.0) { t = 1.0 / t; flip = true; }
  double base = 0.0;
  if (t > T8) { t = (t - 1.0) / (t + 1.0); base = PI_4; }
  double z = t * t;
  double s = 0.0;
#pragma unroll
  for (int k = 23; k >= 0; --k) {
    double c = 1.0 / (double)(2 * k + 1);
    if (k & 1) c = -c;
    s = s * z + c;
  }
  double a = base + t * s;
  if (flip) a = PI_2 - a;
  return a;
}

__device__ __forceinline__ int32_t rcc_sat_rint(double v)
{
  double r = rint(v);
  if (!(r > -2147483648.0)) return INT32_MIN;
  if (r > 2147483647.0) return INT32_MAX;
  return (int32_t)r;
}

__device__ __forceinline__ void rcc_map_q5(const rcc_cam& c, int u, int v, int32_t& X, int32_t& Y)
{
  double x = ((double)u - c.cx) / c.fx;
  double y = ((double)v - c.cy) / c.fy;
  double xs, ys;
  if (c.model == RCC_DIST_PLUMB_BOB) {
    const double k1 = c.D[0], k2 = c.D[1], p1 = c.D[2], p2 = c.D[3], k3 = c.D[4];
    double x2 = x * x, y2 = y * y;
    double r2 = x2 + y2;
    double _2xy = (2.0 * x) * y;
    double kr = k3 * r2;
    kr = kr + k2;
    kr = kr * r2;
    kr = kr + k1;
    kr = kr * r2;
    kr = 1.0 + kr;
    double tx = 2.0 * x2;
    tx = r2 + tx;
    double ty = 2.0 * y2;
    ty = r2 + ty;
    double xd = x * kr;
    double a = p1 * _2xy;
    xd = xd + a;
    a = p2 * tx;
    xd = xd + a;
    double yd = y * kr;
    a = p1 * ty;
    yd = yd + a;
    a = p2 * _2xy;
    yd = yd + a;
    xs = c.fx * xd;
    xs = xs + c.cx;
    ys = c.fy * yd;
    ys = ys + c.cy;
  } else if (c.model == RCC_DIST_FISHEYE) {
    const double k1 = c.D[0], k2 = c.D[1], k3 = c.D[2], k4 = c.D[3];
    double x2 = x * x, y2 = y * y;
    double r = sqrt(x2 + y2);
    double th = rcc_atan_pos(r);
    double t2 = th * th;
    double p = k4 * t2;
    p = p + k3;
    p = p * t2;
    p = p + k2;
    p = p * t2;
    p = p + k1;
    p = p * t2;
    p = 1.0 + p;
    double thd = th * p;
    double s = (r > 1e-8) ? thd / r : 1.0;
    xs = c.fx * x;
    xs = xs * s;
    xs = xs + c.cx;
    ys = c.fy * y;
    ys = ys * s;
    ys = ys + c.cy;
  } else {
    xs = (double)u;
    ys = (double)v;
  }
  X = rcc_sat_rint(xs * 32.0);
  Y = rcc_sat_rint(ys * 32.0);
}

__device__ __forceinline__ int rcc_grey_of(int b, int g, int r)
{
  return (b * 1868 + g * 9617 + r * 4899 + 8192) >> 14;
}

// RGB = true: RCC_PIX_RGB8 (sensor_msgs "rgb8": byte 0 is red) -- the same luma with the two outer weights exchanged
template <int NCH, bool RGB = false>
__device__ __forceinline__ int rcc_tap(const uint8_t* __restrict__ src, int stride, int w, int h, int ix, int iy)
{
  if ((unsigned)ix >= (unsigned)w || (unsigned)iy >= (unsigned)h) return 0;
  const uint8_t* p = src + (size_t)iy * stride + (size_t)ix * NCH;
  if (NCH == 3) return RGB ? rcc_grey_of(p[2], p[1], p[0]) : rcc_grey_of(p[0], p[1], p[2]);
  return p[0];
}

// ---- variant 1: staged -----------------------------------------------------------------------
// Destination tile 128 x 8 (256 threads x 4 pixels).  The block computes the bounding box of the
// source pixels its tile touches (the map is the same for every frame), aligns it to 16-pixel
// groups, and for every frame of its group: loads the box with 16 B/lane coalesced loads (48 B =
// 16 BGR pixels per lane), converts each source pixel to grey ONCE, keeps the grey box in LDS
// (double-buffered: the loads of frame f+1 are in flight while frame f's taps are taken), and reads
// the four bilinear taps from LDS.  Out-of-image groups are zero (BORDER_CONSTANT 0).
// Needs width % 128 == 0, height % 8 == 0 and 16 B-aligned rows; a block whose box does not fit the LDS buffer
// (strong magnification) takes the gather path for its tile.
#define ST_TW 128
#define ST_TH 8
#define ST_PITCH 256          // grey bytes per LDS row (>= aligned box width)
#define ST_ROWS 24            // LDS rows per buffer
#define ST_MAXG ((ST_PITCH / 16) * ST_ROWS)
#define ST_SLOTS 3            // 4-pixel source units per thread (box of up to 768 units = 3072 source pixels)

// 16 BGR pixels (48 B in three 16-B registers) -> 16 grey bytes.  grey = (1868 B + 9617 G + 4899 R + 8192) >> 14
// evaluated exactly with byte dot products: each weight w = 64*(w >> 6) + (w & 63), so
//   S = (dot4(px, w >> 6) << 8) + dot4(px, 4*(w & 63)) + 32768 = 4 * (sum + 8192) < 2^24
// and the grey value is byte 2 of S (bits 23:16 = (sum + 8192) >> 14).  17 instructions per 4 pixels.
template <bool RGB = false>
__device__ __forceinline__ uint32_t rcc_grey4(uint32_t d0, uint32_t d1, uint32_t d2)
{
  // bytes: d0 = B0 G0 R0 B1, d1 = G1 R1 B2 G2, d2 = R2 B3 G3 R3   (RGB: R and B exchanged, and so are their weights)
  const uint32_t WHI = RGB ? (76u | (150u << 8) | (29u << 16)) : (29u | (150u << 8) | (76u << 16));            // w >> 6 for B, G, R
  const uint32_t WLO = RGB ? (140u | (68u << 8) | (48u << 16)) : (48u | (68u << 8) | (140u << 16));            // 4 * (w & 63)
  const uint32_t p1 = __builtin_amdgcn_alignbyte(d1, d0, 3);       // B1 G1 R1 .
  const uint32_t p2 = __builtin_amdgcn_alignbyte(d2, d1, 2);       // B2 G2 R2 .
  const uint32_t s0 = (__builtin_amdgcn_udot4(d0, WHI, 0u, false) << 8) + __builtin_amdgcn_udot4(d0, WLO, 32768u, false);
  const uint32_t s1 = (__builtin_amdgcn_udot4(p1, WHI, 0u, false) << 8) + __builtin_amdgcn_udot4(p1, WLO, 32768u, false);
  const uint32_t s2 = (__builtin_amdgcn_udot4(p2, WHI, 0u, false) << 8) + __builtin_amdgcn_udot4(p2, WLO, 32768u, false);
  const uint32_t s3 = (__builtin_amdgcn_udot4(d2, WHI << 8, 0u, false) << 8) + __builtin_amdgcn_udot4(d2, WLO << 8, 32768u, false);
  const uint32_t lo = __builtin_amdgcn_perm(s1, s0, 0x0C0C0602u);  // (s0.b2, s1.b2, 0, 0)
  const uint32_t hi = __builtin_amdgcn_perm(s3, s2, 0x06020C0Cu);  // (0, 0, s2.b2, s3.b2)
  return lo | hi;
}

template <bool RGB = false>
__device__ __forceinline__ uint4 rcc_grey16(const uint4& a, const uint4& b, const uint4& d)
{
  return make_uint4(rcc_grey4<RGB>(a.x, a.y, a.z), rcc_grey4<RGB>(a.w, b.x, b.y), rcc_grey4<RGB>(b.z, b.w, d.x), rcc_grey4<RGB>(d.y, d.z, d.w));
}

#define ST_TILE_LDS (2 * (ST_ROWS * ST_PITCH + 16) + 64)   // two grey buffers + the bounding-box scratch (128 x 8 tile, 256 threads)
// the same for a 128 x TH tile worked by 32 * TH threads (TH = 8 or 16): TH + 16 LDS rows per buffer, one bounding-box slot per wave
#define ST_TILE_LDS_T(TH) (2 * (((TH) + 16) * ST_PITCH + 16) + 16 * ((TH) / 2))

// TH = 16 (round 4): a 128 x 16 tile by 512 threads.  The source box of a tile is its destination rows plus what the lens bends
// into them plus one row of taps: about 10 source rows for 8 destination rows, about 18-19 for 16 -- 1.16 instead of 1.25 source
// rows loaded and converted per destination row, and half the barriers per pixel.  Same arithmetic, same bytes.
// One 128 x TH destination tile by 32 * TH threads (tid), frames [bz * fpb, ...).  `lds`: this tile's ST_TILE_LDS bytes;
// `s_flag`: one int per tile of the workgroup (a workgroup may run two tiles side by side: `half`, `nhalves`); every
// thread of the workgroup must call this the same number of times -- the barriers inside are workgroup barriers, and
// the "box does not fit LDS" fallback is taken by all tiles of the workgroup together.  tile < 0: no tile (idle half).
template <int NCH, bool RGB = false, int TH = ST_TH>
__device__ __forceinline__ void ingest_staged_body(const uint8_t* __restrict__ frames,
                                                   int64_t frame_bytes, int stride, int w, int h,
                                                   const rcc_cam& cam, uint8_t* __restrict__ grey,
                                                   int nframes, int fpb, int ntx, const int tile, const int bz, const int tid,
                                                   uint8_t* lds, int* s_flag, const int half, const int nhalves,
                                                   const int2* __restrict__ map = nullptr, const int4* __restrict__ tilebox = nullptr)
{
  static_assert(TH == 8 || TH == 16 || TH == 32, "tile height");
  constexpr int NT = 32 * TH;                        // threads of the tile
  constexpr int ROWS = TH + 16;                      // LDS rows per buffer (ST_ROWS for TH = 8)
  constexpr int SLOTS = (TH == 8) ? ST_SLOTS : 2;    // 4-pixel source units per thread: boxes of up to 768 / 1024 units
  const int tv = tile < 0 ? 0 : tile;
  const int by = tv / ntx, bx = tv - by * ntx;
  uint8_t (*sbuf)[ROWS * ST_PITCH + 16] = reinterpret_cast<uint8_t (*)[ROWS * ST_PITCH + 16]>(lds);   // two buffers; +16: dump slot of idle lanes
  int (*s_red)[4] = reinterpret_cast<int (*)[4]>(lds + 2 * (ROWS * ST_PITCH + 16));                       // [NT / 64][4]
  const int tx = tid & 31, ty = tid >> 5;           // 32 quads x TH rows
  const int x0 = bx * ST_TW + tx * 4;
  const int y = by * TH + ty;
  const bool inside = (tile >= 0) && (y < h) && (x0 < w);          // w % 16 == 0: a quad is all in or all out
  int32_t X[4], Y[4];
  int mnx = INT32_MAX, mxx = INT32_MIN, mny = INT32_MAX, mxy = INT32_MIN;
  if (map) {
    // the Q5 map and the tile's source box are frame-invariant: tabulated once per handle by k_ingest_map (the same
    // rcc_map_q5, so the same bits) -- 8 B per destination pixel per block of `fpb` frames instead of ~100 fp64
    // instructions per pixel and a block-wide reduction (the pass is bound by instruction issue, not by HBM)
#pragma unroll
    for (int j = 0; j < 4; ++j) { X[j] = 0; Y[j] = 0; }
    if (inside) {
      const int4* mp = reinterpret_cast<const int4*>(map + (size_t)y * w + x0);
      const int4 a = mp[0], b = mp[1];
      X[0] = a.x; Y[0] = a.y; X[1] = a.z; Y[1] = a.w; X[2] = b.x; Y[2] = b.y; X[3] = b.z; Y[3] = b.w;
    }
    const int4 tb = tilebox[tv];
    mnx = tb.x; mxx = tb.y; mny = tb.z; mxy = tb.w;
  } else {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    X[j] = 0; Y[j] = 0;
    if (inside) {
      rcc_map_q5(cam, x0 + j, y, X[j], Y[j]);
      mnx = min(mnx, X[j] >> 5); mxx = max(mxx, X[j] >> 5);
      mny = min(mny, Y[j] >> 5); mxy = max(mxy, Y[j] >> 5);
    }
  }
  // block-wide bounding box
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    mnx = min(mnx, __shfl_xor(mnx, off, 64)); mxx = max(mxx, __shfl_xor(mxx, off, 64));
    mny = min(mny, __shfl_xor(mny, off, 64)); mxy = max(mxy, __shfl_xor(mxy, off, 64));
  }
  if ((tid & 63) == 0) { s_red[tid >> 6][0] = mnx; s_red[tid >> 6][1] = mxx; s_red[tid >> 6][2] = mny; s_red[tid >> 6][3] = mxy; }
  __syncthreads();
  mnx = s_red[0][0]; mxx = s_red[0][1]; mny = s_red[0][2]; mxy = s_red[0][3];
#pragma unroll
  for (int q = 1; q < NT / 64; ++q) { mnx = min(mnx, s_red[q][0]); mxx = max(mxx, s_red[q][1]); mny = min(mny, s_red[q][2]); mxy = max(mxy, s_red[q][3]); }
  }
  const int f0 = bz * fpb;
  const int f1 = min(f0 + fpb, nframes);
  // box in source pixels: columns [bxa, bxa + 4*upr), rows [by0, by0 + bh); taps need +1.  The box starts on a 4-pixel
  // unit (one grey dword in LDS, 12 bytes of BGR: dword-aligned loads), not on a 16-pixel group: ~35 instead of 40 units
  // per row of a 128-pixel tile -- an eighth fewer loads and conversions
  const long long spanx = (long long)mxx - (long long)mnx, spany = (long long)mxy - (long long)mny;
  const int bxa = (int)((unsigned)mnx & ~3u);       // arithmetic: floor to a multiple of 4 (two's complement)
  const int upr = (spanx < 100000) ? (((mxx + 1) - bxa) / 4 + 1) : (1 << 20);        // units per box row
  const int bh = (spany < 100000) ? (mxy + 1 - mny + 1) : (1 << 20);
  const int by0 = mny;
  const bool fits_here = (tile >= 0) && (upr * 4 <= ST_PITCH) && (bh <= ROWS) && (upr * bh <= NT * SLOTS);   // uniform over the tile's threads
  if (tid == 0) s_flag[half] = fits_here ? 1 : 0;
  __syncthreads();
  const bool fits = s_flag[0] && (nhalves == 1 || s_flag[1]);      // workgroup-uniform: the barriers below need every thread

  if (!fits) {
    // gather path for this tile (same arithmetic, taps from global memory)
    if (!inside) return;
    for (int f = f0; f < f1; ++f) {
      const uint8_t* src = frames + (size_t)f * frame_bytes;
      uint32_t out = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int ix = X[j] >> 5, iy = Y[j] >> 5, fx = X[j] & 31, fy = Y[j] & 31;
        int p00 = rcc_tap<NCH, RGB>(src, stride, w, h, ix, iy), p01 = rcc_tap<NCH, RGB>(src, stride, w, h, ix + 1, iy);
        int p10 = rcc_tap<NCH, RGB>(src, stride, w, h, ix, iy + 1), p11 = rcc_tap<NCH, RGB>(src, stride, w, h, ix + 1, iy + 1);
        int acc = (32 - fx) * (32 - fy) * p00 + fx * (32 - fy) * p01 + (32 - fx) * fy * p10 + fx * fy * p11;
        out |= (uint32_t)((acc + 512) >> 10) << (8 * j);
      }
      *reinterpret_cast<uint32_t*>(grey + (size_t)f * w * h + (size_t)y * w + x0) = out;
    }
    return;
  }

  // The source box is dealt out in 4-pixel units (12 B of BGR = one grey dword) over ALL threads of the block,
  // unit u = tid + 256 * slot: every wave loads and converts its share (typically 1.5 units per thread), so the
  // conversion no longer sits on the one or two waves that would own whole 16-pixel groups -- the block's waves
  // reach the barrier together.
  const int nunits = upr * bh;
  int uoff[SLOTS], ulds[SLOTS];
  bool uact[SLOTS], uin[SLOTS];
#pragma unroll
  for (int sl = 0; sl < SLOTS; ++sl) {
    const int u = ((tid + 64 * (tv & 3)) & (NT - 1)) + NT * sl;     // the waves that get the partly filled last slot rotate with the tile
    const int ur = u / upr, uc = u - ur * upr;
    uact[sl] = u < nunits;
    const int sx = bxa + 4 * uc, sy = by0 + ur;
    uin[sl] = uact[sl] && sx >= 0 && sx < w && sy >= 0 && sy < h;
    uoff[sl] = min(max(sy, 0), h - 1) * stride + min(max(sx, 0), w - 4) * NCH;     // clamped: always a valid address
    ulds[sl] = uact[sl] ? ur * ST_PITCH + 4 * uc : ROWS * ST_PITCH;             // idle units write the dump slot
  }
  // per destination pixel: LDS dword address of the tap pair, byte shift, and the bilinear weights
  // in the form the byte dot product takes them
  int taddr[4], tsh[4], wx[4], wy0[4], wy1[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    // (a thread below the image -- the lower half of a 16-row tile on the last tile row of a height that is 8 mod 16 -- reads
    // the box's first dword and stores nothing)
    const int toff = inside ? ((Y[j] >> 5) - by0) * ST_PITCH + ((X[j] >> 5) - bxa) : 0;
    const int fx = X[j] & 31, fy = Y[j] & 31;
    taddr[j] = toff & ~3;
    tsh[j] = toff & 3;
    wx[j] = (32 - fx) | (fx << 8);
    wy0[j] = 64 * (32 - fy);
    wy1[j] = 64 * fy;
  }

  // Straight-line frame loop (the variant requires width % 128 == 0 and height % 8 == 0, so every
  // thread inside the image owns 4 destination pixels): loads are unconditional within an active slot -- an out-of-image unit reads
  // a clamped in-image address and is zeroed by a select (BORDER_CONSTANT 0) -- so the compiler can count vmcnt
  // instead of draining it.  Buffer addressing: a per-frame descriptor (scalar arithmetic) + the thread's constant
  // 32-bit offsets.
  typedef unsigned u32x3_t __attribute__((ext_vector_type(3)));
  struct Regs { u32x3_t q[SLOTS]; };
  bool slot_any[SLOTS];
#pragma unroll
  for (int sl = 0; sl < SLOTS; ++sl) slot_any[sl] = __any(uact[sl]);     // wave-uniform
  auto issue = [&](int f, Regs& r) {
    const int fc = min(f, f1 - 1);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(frames + (size_t)fc * frame_bytes), 0, (int)frame_bytes, 0x00020000);
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
      if (!slot_any[sl]) continue;
      if (NCH == 3) r.q[sl] = __builtin_amdgcn_raw_buffer_load_b96(rs, uoff[sl], 0, 0);
      else r.q[sl].x = __builtin_amdgcn_raw_buffer_load_b32(rs, uoff[sl], 0, 0);
    }
  };
  auto commit = [&](uint8_t* buf, const Regs& r) {
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
      if (!slot_any[sl]) continue;
      uint32_t g4 = (NCH == 3) ? rcc_grey4<RGB>(r.q[sl].x, r.q[sl].y, r.q[sl].z) : r.q[sl].x;
      if (!uin[sl]) g4 = 0;
      *reinterpret_cast<uint32_t*>(buf + ulds[sl]) = g4;
    }
  };
  const int out_off = inside ? y * w + x0 : 0x7FFFFFF0;          // beyond the descriptor's range: the hardware drops the store
  auto taps = [&](int f, const uint8_t* L) {
    uint32_t sv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // aligned dword pairs + v_alignbyte: a misaligned ds_read_u16 serialises in the LDS
      // (measured: ~48 LDS cycles per instruction, SQ_WAIT_INST_LDS = 51 % of wave time)
      const uint32_t* t = reinterpret_cast<const uint32_t*>(L + taddr[j]);
      const uint32_t top = __builtin_amdgcn_alignbyte(t[1], t[0], (uint32_t)tsh[j]);                              // p00 p01 . .
      const uint32_t bot = __builtin_amdgcn_alignbyte(t[ST_PITCH / 4 + 1], t[ST_PITCH / 4], (uint32_t)tsh[j]);    // p10 p11 . .
      // acc = (32-fx)(32-fy) p00 + fx (32-fy) p01 + (32-fx) fy p10 + fx fy p11, rows first; 64*(acc+512) has
      // (acc+512) >> 10 in byte 2
      const uint32_t th = __builtin_amdgcn_udot4(top, (uint32_t)wx[j], 0u, false);
      const uint32_t bh = __builtin_amdgcn_udot4(bot, (uint32_t)wx[j], 0u, false);
      sv[j] = __umul24(th, (uint32_t)wy0[j]) + (__umul24(bh, (uint32_t)wy1[j]) + 32768u);
    }
    const uint32_t lo = __builtin_amdgcn_perm(sv[1], sv[0], 0x0C0C0602u);
    const uint32_t hi = __builtin_amdgcn_perm(sv[3], sv[2], 0x06020C0Cu);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(grey + (size_t)f * w * h, 0, w * h, 0x00020000);
    // non-temporal: the 2 GB of grey a batch writes are read again only by the next kernel, long after they have left
    // every cache; measured +1 % frames/s (the ingest pass 1.73 -> 1.70 ms)
    __builtin_amdgcn_raw_buffer_store_b32(lo | hi, ro, out_off, 0, 2 /* nt */);
  };

  // software pipeline: at step f the loads of f+2 are issued, the taps of f are taken from LDS buffer f&1, and frame
  // f+1 (loaded two steps ago) is converted into buffer (f+1)&1.  Unrolled by two so the register sets keep static names.
  // (The loads sit behind wave-uniform branches -- slot_any -- so hipcc cannot count them and waits with vmcnt(1) before
  // each conversion, i.e. also for the loads just issued.  Instantiating the loop per slot count makes the counts static
  // (vmcnt(2..6)) and costs 8 registers: measured the same 1.67 ms per 1024 x 1080p -- the pass is not short of loads in
  // flight, its waves wait for an issue slot half of their life (SQ_WAIT_INST_ANY) -- so the simple form stays.)
  uint8_t* const b0 = sbuf[0];
  uint8_t* const b1 = sbuf[1];
  Regs r0, r1;
  issue(f0, r0);
  issue(f0 + 1, r1);
  commit(b0, r0);
  for (int f = f0; f < f1; f += 2) {
    __syncthreads();            // buffer 0 holds frame f; buffer 1 is free
    issue(f + 2, r0);
    __builtin_amdgcn_sched_barrier(0);    // keep the prefetch ahead of the taps and of the conversion
    taps(f, b0);
    __builtin_amdgcn_sched_barrier(0);
    commit(b1, r1);
    __syncthreads();            // buffer 1 holds frame f+1 (if any); buffer 0 is free
    issue(f + 3, r1);
    __builtin_amdgcn_sched_barrier(0);
    if (f + 1 < f1) taps(f + 1, b1);      // block-uniform
    __builtin_amdgcn_sched_barrier(0);
    commit(b0, r0);
  }
}

