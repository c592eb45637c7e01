// k_subpix.hip -- stage a5: sub-pixel corner refinement, cornerSubPix form (SURVEY.md appendix
// B.5), one 64-lane wavefront per corner.
//
// What the reference receives from this stage: pixel_corners_x/y[0..3] of each detection
// (real_preprocessing/src/corner_detections.cpp:53-54, where it casts them to int).
//
// Numerics: binary64 throughout; the five window sums are accumulated
// one bin per lane (bin = sample index mod 64, increasing index) and combined along the xor-butterfly's
// pairing tree (partners l ^ 32, then l ^ 16, .. l ^ 1; own + partner's).  IEEE addition is commutative,
// so the result equals the specification's 64-bin pairwise tree bit for bit.
// Compiled with -ffp-contract=off: every operation below is one rounded IEEE operation.
#include "rcc_internal.h"
#include "wave_reduce.h"

#define SP_MAXW 7
#define SP_MAXP (2 * SP_MAXW + 3)

// Board scenes (FID = false): grid (max_kept, nframes), the wave handles candidate blockIdx.x of frame blockIdx.y.  (Several
// corners per wave, to share the table setup, measured slower there once the tables came from rcc_create: the longer chain
// per wave costs more in the launch's tail than the ~15 saved loads.)  Tag scenes (FID = true): grid (qstep, nframes) with
// qstep well below the list's length; a wave walks its frame's list four candidates at a time -- see below and
// rcc_launch_subpix.
// a5's gate for board scenes: 16 samples on a radius-11 ring
// around the candidate's own pixel against the ring's mid level
__constant__ int8_t c_ring11[16][2] = {
  {11, 0}, {10, 4}, { 8, 8}, { 4,10}, { 0,11}, {-4,10}, {-8, 8}, {-10, 4},
  {-11, 0}, {-10,-4}, {-8,-8}, {-4,-10}, { 0,-11}, { 4,-10}, { 8,-8}, {10,-4}
};

// a5's gate for board scenes as a function of its own -- NOT inlined on purpose: inlined into the candidate loop it raised the kernel from 78
// to 94 registers (five waves per SIMD instead of six: 0.175 -> 0.195 ms per 1024 frames, profiles r04_b / r04_c); a call costs a few
// dozen cycles per candidate.  Returns true (and writes (-1, -1)) when candidate q of frame f cannot be a junction.
template <bool INL>
__device__ __forceinline__ bool subpix_gate_body(const uint8_t* g, const rcc_cand* pre, double* pre_xy, int f, int kstride, int w, int h, int gate_contrast, int lane, const int q) {
    const rcc_cand c0 = pre[(size_t)f * kstride + q];
    const int xi = c0.x, yi = c0.y;
    if (!(xi >= 11 && yi >= 11 && xi < w - 11 && yi < h - 11)) return false;      // wave-uniform
    const int k = lane & 15;
    const int v = (int)g[(size_t)(yi + c_ring11[k][1]) * w + (xi + c_ring11[k][0])];
    int lo = v, hi = v;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) { lo = min(lo, __shfl_xor(lo, off, 64)); hi = max(hi, __shfl_xor(hi, off, 64)); }
    const unsigned bits = (unsigned)__builtin_amdgcn_ballot_w64(v > ((lo + hi) >> 1)) & 0xFFFFu;
    const unsigned rotl = ((bits << 1) | (bits >> 15)) & 0xFFFFu;
    const int lo0 = __builtin_amdgcn_readfirstlane(lo), hi0 = __builtin_amdgcn_readfirstlane(hi);
    if (hi0 - lo0 >= gate_contrast && __popc(bits ^ rotl) >= 4) return false;
    if (lane == 0) {
      pre_xy[((size_t)f * kstride + q) * 2] = -1.0;
      pre_xy[((size_t)f * kstride + q) * 2 + 1] = -1.0;
    }
    return true;
}
__device__ __attribute__((noinline)) bool subpix_gated(const uint8_t* g, const rcc_cand* pre, double* pre_xy, int f, int kstride, int w, int h, int gate_contrast, int lane, const int q)
{
  return subpix_gate_body<false>(g, pre, pre_xy, f, kstride, w, h, gate_contrast, lane, q);
}

template <bool FID>       // FID: tag scenes (the convex-black-corner test in front of the refinement, waves walk the candidate list)
__global__ __launch_bounds__(64) void k_subpix(const uint8_t* __restrict__ grey, int w, int h,
                                               const rcc_cand* __restrict__ pre, const int32_t* __restrict__ npre,
                                               rcc_subpix_params sp, const rcc_subpix_lane* __restrict__ tab, int kstride,
                                               double* __restrict__ pre_xy, int gate_contrast /* min_contrast of the gate in front of the refinement; < 0: no gate (board scenes) */, int qstep)
{
  __shared__ double S[SP_MAXP * SP_MAXP];
  const int f = blockIdx.y;
  const int np = npre[f];
  if ((int)blockIdx.x >= np) return;
  const int lane = threadIdx.x;
  const uint8_t* g = grey + (size_t)f * w * h;
  // Board scenes: is candidate q worth refining?  The 36 L-shaped corners on the outline of a 9 x 7-square board took 42 % of this
  // stage's iterations only to be rejected by a4.3, and a cluttered scene brings hundreds of such candidates.  Lanes 0..15 read the
  // radius-11 ring around the candidate's pixel (a Harris maximum sits up to ~3 px off its junction, 6.5 px under heavy blur: the
  // ring still encloses it); fewer than four transitions against the ring's own mid level, or a ring that does not span
  // min_contrast: not a junction -- the wave writes (-1, -1), which a4.3 rejects.  A ring that leaves the image passes.
  auto gated = [&](const int q) -> bool { return subpix_gated(g, pre, pre_xy, f, kstride, w, h, gate_contrast, lane, q); };
  // the wave's FIRST candidate is tested in line, in front of everything (nothing else is live yet): in a board frame without clutter
  // it is the only one, and most waves end here or go straight on to their one refinement; only the walk over a longer list calls
  const bool gate_on = !FID && gate_contrast >= 0;
  bool first_gated = false;
  if (gate_on) {
    first_gated = subpix_gate_body<true>(g, pre, pre_xy, f, kstride, w, h, gate_contrast, lane, (int)blockIdx.x);
    if (first_gated && (int)blockIdx.x + qstep >= np) return;
  }
  const int win = sp.win;
  const int ww = 2 * win + 1, pw = 2 * win + 3;
  // per-lane sample tables (the window geometry does not change between corners or iterations): patch samples
  // idx = lane + 64 t < pw^2 and gradient samples k = lane + 64 t < ww^2, tabulated at rcc_create -- no integer division here
  constexpr int PT = RCC_SP_PT, GT = RCC_SP_GT;
  static_assert(PT == (SP_MAXP * SP_MAXP + 63) / 64 && GT == ((2 * SP_MAXW + 1) * (2 * SP_MAXW + 1) + 63) / 64, "table shape");
  int poff[PT];            // (i - win - 1) * w + (j - win - 1): offset of the sample's top-left tap from (iy, ix)
  int goff[GT];            // (i + 1) * pw + (j + 1): centre of the gradient stencil in S
  double gm[GT], gpx[GT], gpy[GT];
  auto load_tables = [&](const int tl) {
    const rcc_subpix_lane T = tab[tl];
#pragma unroll
    for (int t = 0; t < PT; ++t) poff[t] = T.poff[t];
#pragma unroll
    for (int t = 0; t < GT; ++t) { goff[t] = T.goff[t]; gm[t] = T.gm[t]; gpx[t] = (double)T.gpx[t]; gpy[t] = (double)T.gpy[t]; }
  };
  // Tag scenes load the tables once, in front of the walk over the list.  Board scenes load them INSIDE the loop, behind the gate: a
  // wave whose candidate is gated (the common case: 36 outline candidates of ~ 85 per frame) retires without them; the asm keeps the
  // compiler from hoisting the loop-invariant load in front of the gate.
  if (FID) load_tables(lane);
  // candidates blockIdx.x, + qstep, ... of the frame (qstep = the grid's width: one candidate per wave where the lists are
  // short; tag scenes hold ~750 candidates of up to 2048, two thirds of which leave at the test below, and a wave that finds
  // its slot unused still costs a slot for a microsecond -- there the grid is narrower than the list and a wave walks it)
  constexpr bool fid = FID;
  const int per_pass = fid ? 4 : 1;
  for (int q0 = blockIdx.x; q0 < np; q0 += per_pass * qstep) {
  if (!FID) {
    if (gate_on && (q0 == (int)blockIdx.x ? first_gated : gated(q0))) continue;
    int tl = lane;
    asm volatile("" : "+v"(tl));
    load_tables(tl);
  }
  unsigned refine_m = 1u;                  // bit k: candidate q0 + k * qstep is refined
  int cgx = 0, cgy = 0;                    // fid: the candidate of this lane's group of 16
  if (fid) {
    // Square fiducials: only what can become a quad corner is refined.  A candidate whose own pixel fails the convex-black-
    // corner test of the quad stage (k_fid.hip fid_corner_class: 16 samples on a radius-5 ring against the ring's mid level,
    // exactly one black arc of 2..7 samples) keeps that pixel as its position: the quad stage's test then fails for it at
    // the same pixel.  About two thirds of a tag scene's candidates (the payload's inner corners) leave here, so the test is
    // made for FOUR candidates per pass -- the four 16-lane groups of the wave, lane l of a group on ring sample l -- and its
    // two dependent reads (candidate, ring pixel) are paid once for the four; with two transitions the black samples form
    // one arc, so the arc's length is their count.
    const int grp = lane >> 4;
    const int myq = q0 + grp * qstep;
    const bool valid = myq < np;
    rcc_cand cg; cg.x = 0; cg.y = 0; cg.score = 0;
    if (valid) cg = pre[(size_t)f * kstride + myq];
    const int xi = cg.x, yi = cg.y;
    cgx = xi; cgy = yi;
    const bool room = valid && xi >= 5 && yi >= 5 && xi < w - 5 && yi < h - 5;        // uniform over the group
    // ring offsets + 5, one nibble per sample: x = 5 5 4 2 0 -2 -4 -5 -5 -5 -4 -2 0 2 4 5, y = the same a quarter turn on
    const int k4 = 4 * (lane & 15);
    const int rx = (int)((0xA9753100013579AAull >> k4) & 15ull) - 5, ry = (int)((0x3100013579AAA975ull >> k4) & 15ull) - 5;
    const int v = room ? (int)g[(size_t)(yi + ry) * w + (xi + rx)] : 0;
    int lo = v, hi = v;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) { lo = min(lo, __shfl_xor(lo, off, 64)); hi = max(hi, __shfl_xor(hi, off, 64)); }
    const int t = (lo + hi) / 2;
    const unsigned long long wb = __builtin_amdgcn_ballot_w64(v > t);
    const unsigned bits = (unsigned)(wb >> (16 * grp)) & 0xFFFFu;                       // bit k: ring sample k of this group's candidate is white
    const unsigned rotl = ((bits << 1) | (bits >> 15)) & 0xFFFFu;
    const int nblack = 16 - __popc(bits);
    const bool corner = room && (hi - lo >= gate_contrast) && (__popc(bits ^ rotl) == 2) && nblack >= 2 && nblack <= 7;
    if (valid && !corner && (lane & 15) == 0) {
      pre_xy[((size_t)f * kstride + myq) * 2] = (double)xi;
      pre_xy[((size_t)f * kstride + myq) * 2 + 1] = (double)yi;
    }
    const unsigned long long cb = __builtin_amdgcn_ballot_w64(corner);
    refine_m = (unsigned)(cb & 1ull) | ((unsigned)((cb >> 16) & 1ull) << 1) | ((unsigned)((cb >> 32) & 1ull) << 2) | ((unsigned)((cb >> 48) & 1ull) << 3);
  }
  for (int kq = 0; kq < per_pass; ++kq) {
  if (!((refine_m >> kq) & 1u)) continue;                           // wave-uniform
  const int q = q0 + kq * qstep;
  double x0, y0;
  if (fid) {
    x0 = (double)__shfl(cgx, 16 * kq, 64); y0 = (double)__shfl(cgy, 16 * kq, 64);
  } else {
    const rcc_cand c0 = pre[(size_t)f * kstride + q];
    x0 = (double)c0.x; y0 = (double)c0.y;
  }
  double cx = x0, cy = y0;
  int iter = 0;
  bool bad = false;
  double err = 0.0;
  do {
    double flx = floor(cx), fly = floor(cy);
    int ix = (int)flx, iy = (int)fly;
    if (ix - win - 1 < 0 || iy - win - 1 < 0 || ix + win + 2 > w - 1 || iy + win + 2 > h - 1) {
      bad = true;
      break;
    }
    double fx = cx - flx, fy = cy - fly;
    double ofx = 1.0 - fx, ofy = 1.0 - fy;
    double a00 = ofx * ofy, a01 = fx * ofy, a10 = ofx * fy, a11 = fx * fy;
    __syncthreads();   // previous iteration's readers are done with S
    const uint8_t* pc = g + (size_t)iy * w + ix;
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      const int idx = lane + 64 * t;
      if (idx < pw * pw) {
        const uint8_t* p = pc + poff[t];
        // the two taps of a row with one (unaligned) 16-bit load
        unsigned short r0, r1;
        __builtin_memcpy(&r0, p, 2);
        __builtin_memcpy(&r1, p + w, 2);
        double t0 = a00 * (double)(r0 & 255);
        double t1 = a01 * (double)(r0 >> 8);
        double t2 = a10 * (double)(r1 & 255);
        double t3 = a11 * (double)(r1 >> 8);
        double s = t0 + t1;
        s = s + t2;
        s = s + t3;
        S[idx] = s;
      }
    }
    __syncthreads();
    double a = 0.0, b = 0.0, c = 0.0, b1 = 0.0, b2 = 0.0;
#pragma unroll
    for (int t = 0; t < GT; ++t) {
      const int k = lane + 64 * t;
      if (k < ww * ww) {
        const double* spp = S + goff[t];
        double gx = spp[1] - spp[-1];
        double gy = spp[pw] - spp[-pw];
        double m = gm[t];
        double gxx = (gx * gx) * m;
        double gxy = (gx * gy) * m;
        double gyy = (gy * gy) * m;
        double px = gpx[t], py = gpy[t];
        a = a + gxx;
        b = b + gxy;
        c = c + gyy;
        double u1 = gxx * px, u2 = gxy * py;
        b1 = b1 + (u1 + u2);
        double v1 = gxy * px, v2 = gyy * py;
        b2 = b2 + (v1 + v2);
      }
    }
    // the five sums at once (wave_reduce.h): per quantity the pairing is the xor butterfly's (lane bit 5 first,
    // bit 0 last; own + partner's, and IEEE addition commutes), so each total is the specification's 64-bin
    // pairwise tree bit for bit; quantity q ends in lanes 8q..8q+7 and is broadcast through scalar registers
    double red[8] = { a, b, c, b1, b2, 0.0, 0.0, 0.0 };
    wred::reduce_scatter<8>(lane, red);
    a = wred::bcast<0>(red[0]);
    b = wred::bcast<8>(red[0]);
    c = wred::bcast<16>(red[0]);
    const double bb1 = wred::bcast<24>(red[0]), bb2 = wred::bcast<32>(red[0]);
    double ac = a * c, bsq = b * b;
    double det = ac - bsq;
    if (fabs(det) <= 2.2204460492503131e-16 * 2.2204460492503131e-16) break;
    double scale = 1.0 / det;
    double cs = c * scale, bs = b * scale, as = a * scale;
    double q1 = cs * bb1, q2 = bs * bb2;
    double dx = q1 - q2;
    double q3 = as * bb2, q4 = bs * bb1;
    double dy = q3 - q4;
    double nx = cx + dx, ny = cy + dy;
    double ex = nx - cx, ey = ny - cy;
    double e1 = ex * ex, e2 = ey * ey;
    err = e1 + e2;
    cx = nx; cy = ny;
    if (cx < 0.0 || cx >= (double)w || cy < 0.0 || cy >= (double)h) break;
  } while (++iter < sp.max_iter && err > sp.eps2);
  if (bad || fabs(cx - x0) > (double)win || fabs(cy - y0) > (double)win) { cx = x0; cy = y0; }
  if (lane == 0) {
    pre_xy[((size_t)f * kstride + q) * 2] = cx;
    pre_xy[((size_t)f * kstride + q) * 2 + 1] = cy;
  }
  }
  }
}

hipError_t rcc_launch_subpix(rcc_handle* h, const uint8_t* d_grey, int nframes, hipStream_t s)
{
  const rcc_config& c = h->cfg;
  if (nframes <= 0) return hipSuccess;
  const bool fid = c.target_kind == RCC_TARGET_FIDUCIAL;
  // board scenes: a grid 256 wide (the bench's frames hold 85-130 candidates); a cluttered frame's longer list (up to 2048) is
  // walked by the same waves
  int max_kept = fid ? (c.max_kept < h->kept_cap ? c.max_kept : h->kept_cap) : RCC_MAX_KEPT;
  // tag scenes: the grid is narrower than the list (measured on 1024 x 1080p with ~750 candidates per frame: width 2048 /
  // 512 / 256 / 128 / 64 / 32 / 16: 0.63 / 0.55 / 0.49 / 0.46 / 0.44 / 0.48 / 0.52 ms) -- at least 64 wide, and wide
  // enough that a small batch still puts four waves on every slot of the device
  int qstep = max_kept;
  if (fid) {
    qstep = (24576 + nframes - 1) / nframes;
    if (qstep < 64) qstep = 64;
    if (h->subpix_grid > 0) qstep = h->subpix_grid;       // rcc_set_subpix_grid
    if (qstep > max_kept) qstep = max_kept;
  }
  if (fid)
    hipLaunchKernelGGL(k_subpix<true>, dim3(qstep, nframes), dim3(64), 0, s, d_grey, c.width, c.height,
                       h->d_pre, h->d_npre, h->sp, h->d_sp_tab, h->kept_cap, h->d_pre_xy, c.thr_min_contrast > 0 ? c.thr_min_contrast : 0, qstep);
  else
    hipLaunchKernelGGL(k_subpix<false>, dim3(qstep, nframes), dim3(64), 0, s, d_grey, c.width, c.height,
                       h->d_pre, h->d_npre, h->sp, h->d_sp_tab, h->kept_cap, h->d_pre_xy,
                       (c.xj_check && c.target_kind == RCC_TARGET_CHECKERBOARD) ? (c.thr_min_contrast > 0 ? c.thr_min_contrast : 0) : -1, qstep);
  return hipGetLastError();
}
