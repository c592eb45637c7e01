// k_dense_wave.hip -- variant 4 of the threshold + corner pass (a3 + a4.1), compact-map form only (what
// rcc_detect_batch runs): ONE INDEPENDENT WAVEFRONT per (frame, segment, band, window), no workgroup barrier.
//
// Same definitions, same per-lane arithmetic (dense_rows.h) and bit-identical outputs as the band kernel
// (k_dense_band.hip).  What changes is the unit of scheduling.  In the band kernel eight windows march in lockstep,
// one s_barrier per tile row, because they share the staged rows: a wave whose window is flat waits at every barrier for
// the waves that run the corner stages, and its slot is held until the whole workgroup retires.  The pass is bound by
// vector-instruction issue, a wave alone issues at ~3/4 of the SIMD's rate, and the active
// windows of a frame are always the same ones -- so what the lockstep costs is SIMDs on which too few waves are ready.
// Here every window is its own 64-thread workgroup with its own 4-KiB ring in LDS:
//   * per tile row the wave issues ONE LDS-DMA load (buffer_load_dwordx4 ... lds: its window's 4 rows x 256 B) and ONE
//     byte store (its 61 tile levels); the corner stages re-read the rows from its ring; the 12 halo pixels a window
//     shares with its neighbour are fetched twice (5 %, from L2);
//   * a flat window runs ahead at the speed of the front stage alone and retires early; the dispatcher refills its slot
//     with the next job, so a SIMD keeps six waves that all have work;
//   * ordering is the wave's own: a counted s_waitcnt vmcnt(2 * WAVE_DEPTH - 1) before a staged tile row is read (every
//     iteration issues exactly one DMA and one store, both unconditional -- a row or lane with nothing to write gets an
//     out-of-range offset, which the hardware drops), no LDS traffic between waves at all.
// The full binary image (stage form) stays with the band kernel: a window's 244-byte spans would be written as partial
// 128-byte lines, one dword per lane and row -- built and measured: 1.55-1.64 ms per 1024 x 1080p with non-temporal
// stores, 1.35 ms with cached ones, against the band kernel's 0.90-0.92 (scratch/membench3.hip had the stores alone at
// 0.58 ms against 0.36 ms for whole lines).
#include <cstdlib>
#include "dense_wave_body.h"

template <int PRIO, int GANG>
__global__ __launch_bounds__(64 * GANG) __attribute__((amdgpu_waves_per_eu(6, 6)))
void k_dense_wave(const uint8_t* __restrict__ grey, int w, int h, int nbands, int nwin, int nseg, int seg_tiles, int nframes,
                  int min_contrast, int hthresh, int margin, int cap, int allow_skip, uint8_t* __restrict__ thr_map,
                  rcc_cand* __restrict__ cand, int32_t* __restrict__ cand_count, int fchunk, int syncmask)
{
  __shared__ __attribute__((aligned(1024))) uint8_t ring_all[GANG * WAVE_RING * 1024];
  const int wib = (GANG > 1) ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
  dense_wave_body<PRIO, GANG>(grey, w, h, nbands, nwin, nseg, seg_tiles, nframes, min_contrast, hthresh, margin, cap, allow_skip, thr_map, cand, cand_count,
                              fchunk, syncmask, blockIdx.x, wib, ring_all + wib * (WAVE_RING * 1024));
}

bool rcc_dense_wave_supported(const rcc_handle* h, const uint8_t* d_grey)
{
  return rcc_dense_band_supported(h, d_grey, nullptr) && h->cfg.width >= 256;
}

int rcc_dense_allow_skip(const rcc_handle* h);

void rcc_dense_wave_plan(const rcc_handle* h, int nframes, rcc_wave_plan* p)
{
  const rcc_config& c = h->cfg;
  p->nbands = (c.width + BAND_W - 1) / BAND_W;
  const int bw = c.width < BAND_W ? c.width : BAND_W;
  p->nwin = (bw + STRIP_USE - 1) / STRIP_USE;
  p->fchunk = nframes >= 1024 ? 16 : nframes >= 64 ? nframes / 64 : 1;
  // Segments: jobs are single waves, so the count only trades the four warm-up tile rows of a segment against the length
  // of the longest chains at the end of the launch.  Measured on 1024 x 1080p (270 tile rows): 2 / 3 / 4 / 5 / 6 / 8 / 9 /
  // 12 segments 0.79 / 0.76 / 0.73 / 0.70 / 0.72 / 0.69 / 0.71 / 0.73 ms -- about 34 tile rows per segment.
  const int th = c.height >> 2;
  int nseg = (th + 17) / 34;
  if (nseg < 1) nseg = 1;
  // a batch too small to fill the device (one frame per call is what the ROS node does): shorter segments, down to 8 tile
  // rows, until the jobs fill the wave slots -- the jobs are dependent chains, and with slots to spare the chain's length is
  // the pass's duration (one 1080p frame: 51 us at 8 segments of 34 tile rows)
  {
    static const int cus = [] { hipDeviceProp_t pr; int d = 0; return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&pr, d) == hipSuccess) ? pr.multiProcessorCount : 256; }();
    const long long slots = 24LL * cus;
    const int nseg_max = th / 8 > 1 ? th / 8 : 1;
    while (nseg < nseg_max && (long long)p->nbands * p->nwin * nseg * nframes < slots) ++nseg;
  }
  p->seg_tiles = (th + nseg - 1) / nseg;
  p->nseg = (th + p->seg_tiles - 1) / p->seg_tiles;
  p->njobs = (long long)p->nbands * p->nseg * p->nwin * ((nframes + 8 * p->fchunk - 1) / (8 * p->fchunk)) * 8 * p->fchunk;
#ifdef RCC_EXPERIMENTS
  static const int deal = getenv("RCC_DENSE_DEAL") ? atoi(getenv("RCC_DENSE_DEAL")) : 0;
#else
  const int deal = 0;
#endif
  if (deal == 1) {            // segment-interleaved deal (dense_wave_body.h): units of one band segment, eight per group
    p->fchunk = 0;
    const long long units = (long long)nframes * p->nseg;
    p->njobs = ((units + 7) / 8) * 8 * (long long)p->nbands * p->nwin;
  }
}

hipError_t rcc_launch_dense_wave(rcc_handle* h, const uint8_t* d_grey, int nframes, rcc_cand* d_cand, int32_t* d_cand_count, hipStream_t s)
{
  const rcc_config& c = h->cfg;
  rcc_wave_plan wp;
  rcc_dense_wave_plan(h, nframes, &wp);
  const int nbands = wp.nbands, nwin = wp.nwin, fchunk = wp.fchunk, nseg = wp.nseg, seg_tiles = wp.seg_tiles;
#ifdef RCC_EXPERIMENTS
  const int th = c.height >> 2;
  if (h->dense_gang_sync > 0 && nwin <= 8) {
    // the gang form (measurement only, librcc_hip_exp.so): one workgroup of eight wavefronts per band segment
    int gs = h->dense_gang_seg > 0 ? h->dense_gang_seg : nseg;
    const int gst = (th + gs - 1) / gs;
    gs = (th + gst - 1) / gst;
    const long long nj = (long long)nbands * gs * ((nframes + 8 * fchunk - 1) / (8 * fchunk)) * 8 * fchunk;
    h->dense_kernel = "k_dense_wave<0, 8>";
    hipLaunchKernelGGL((k_dense_wave<WAVE_PRIO, 8>), dim3((unsigned)nj), dim3(512), 0, s, d_grey, c.width, c.height, nbands, nwin, gs, gst,
                       nframes, c.thr_min_contrast, c.harris_thresh, c.cand_margin, c.max_candidates, rcc_dense_allow_skip(h), h->d_thr,
                       d_cand, d_cand_count, fchunk, h->dense_gang_sync - 1);
    return hipGetLastError();
  }
#endif
  const long long njobs = wp.njobs;
  h->dense_kernel = "k_dense_wave<0, 1>";
  hipLaunchKernelGGL((k_dense_wave<WAVE_PRIO, 1>), dim3((unsigned)njobs), dim3(64), 0, s, d_grey, c.width, c.height, nbands, nwin, nseg, seg_tiles,
                     nframes, c.thr_min_contrast, c.harris_thresh, c.cand_margin, c.max_candidates, rcc_dense_allow_skip(h), h->d_thr,
                     d_cand, d_cand_count, fchunk, h->dense_fmod);
  return hipGetLastError();
}
