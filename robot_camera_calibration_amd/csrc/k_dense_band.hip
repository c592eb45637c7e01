// k_dense_band.hip -- variant 1 (default) of the threshold + corner pass (a3 + a4.1): one 8-wave workgroup
// marches a full-width band of the frame, grey rows staged through an LDS ring by LDS-DMA, binary rows
// staged through LDS so that every global store is whole 128-byte lines.
//
// Same definitions and bit-exact same outputs as k_dense_lds (k_dense.hip) and k_dense_march
// (k_dense_fast.hip); the per-lane arithmetic IS k_dense_march's (dense_rows.h).  What changes is how bytes
// move, because that is what bounded the strip kernel (profiles/r01_g_*):
//   * strip kernel: 12 dword VMEM instructions per wave per tile row (4 loads, 4 re-reads, 4 stores) kept the
//     texture-address FIFO full ~40 % of the time (SQ_VMEM_TA_ADDR_FIFO_FULL), and its 244-byte output spans
//     wrote partial 128-B lines: 0.58 ms per 2.1 GB against 0.36 ms for whole lines (scratch/membench3.hip);
//   * here a wave issues per tile row ONE 1-KiB LDS-DMA load (buffer_load_dwordx4 ... lds: 4 rows x 256 B of
//     its aligned chunk, no VGPRs) and ONE 1-KiB store (4 rows x 256 B, whole lines), the back stage re-reads
//     its rows from LDS instead of L2, and the grey tile buffers leave the register file.
// Layout:
//   * band = output columns [1920 b, 1920 b + 1920) (15 lines); staged image = columns [XI0, XI0 + 2304),
//     XI0 = band start - 128 (0 for the first band); LDS ring slot = one tile row (4 image rows) stored
//     chunk-major [chunk 0..8][row 0..3][256 B], so that one DMA instruction (64 lanes x 16 B, linear in LDS)
//     fills one chunk: lanes 16k..16k+15 fetch row k;
//   * wave j computes window j (pixels band start + 244 j - 8 .. + 256, lanes 2..62 useful) exactly as a strip
//     of the strip kernel does, reading its dwords from the ring at the window's (unaligned) offset;
//   * threshold dwords go to an LDS row buffer; one iteration later wave j stores columns [256 j, 256 j + 256)
//     of the four rows with one dwordx4 store (quarter-wave per row).
// Synchronisation: one s_barrier per tile row.  Ring of BAND_DEPTH + 3 slots, DMA BAND_DEPTH = 2 tile rows ahead
// (1, 2 and 3 measure the same: the kernel is VALU-bound); the wait before the barrier is a COUNTED
// s_waitcnt vmcnt(2 * BAND_DEPTH - 1): every wave issues at least one DMA and exactly one store per iteration (both
// unconditional -- masked lanes / rows get an out-of-range buffer offset, which the hardware drops), so the DMA of
// tile row t has at least 2 * BAND_DEPTH - 1 younger operations when it is waited for (the first tile rows are
// waited for outright in the prologue).  hipcc is kept out of this bookkeeping: the DMA is inline asm, since for
// the builtin it drains vmcnt(0) before every LDS read.
#include "dense_band_body.h"

#define BAND_ARGS const uint8_t* __restrict__ grey, int w, int h, int nbands, int nseg, int seg_tiles, int nframes, int min_contrast, \
                  int hthresh, int margin, int cap, int allow_skip, uint8_t* __restrict__ bin, rcc_cand* __restrict__ cand, int32_t* __restrict__ cand_count, int fchunk, \
                  unsigned long long* __restrict__ flat, int flat_tp
#define BAND_PASS grey, w, h, nbands, nseg, seg_tiles, nframes, min_contrast, hthresh, margin, cap, allow_skip, bin, cand, cand_count, job, lds, flat, flat_tp
// Workgroup -> job.  The hardware deals workgroup ids round-robin over the 8 XCDs, and jobs are (frame, segment, band) with
// the band fastest: with job = workgroup id an XCD always gets the same segment of the same residue class of frames (2
// segments: even XCDs the top halves, odd XCDs the bottom halves), and a batch whose activity has a period that shares a
// factor with 8 keeps part of the chip idle (scratch/t_dense_flat.py, 1024 x 1080p, compact form, one box: every 4th frame
// active 1.09 ms against 0.77 ms with the deal below; 32 scenes repeated 1.11 against 0.98).  So an XCD takes whole
// frames, in chunks of `fchunk` consecutive frames (16 for a batch of 1024), and the chunks go to the XCDs along a diagonal:
// chunk c to XCD (c + c / 8) mod 8 -- one chunk of every group of 8 per XCD, shifted by one from group to group.  A target
// that shows up half way through the batch still loads every XCD (one contiguous eighth of the batch per XCD: 0.99 ms
// against 0.82), no short period lands on one XCD, and consecutive frames on an XCD measure 5 % faster than single frames
// dealt around (0.98 against 1.04 ms on the uniform batch).
// Workgroup id -> (XCD x = id & 7, k = id >> 3); the XCD's k-th job is job k mod (fchunk jpf) of its chunk in group k / (fchunk jpf).
#define BAND_JOB                                                                        \
  const int jpf_ = nbands * nseg, cj_ = fchunk * jpf_, k_ = (int)(blockIdx.x >> 3);     \
  const int grp_ = k_ / cj_, r_ = k_ - grp_ * cj_;                                      \
  const int fr_ = (8 * grp_ + (((int)(blockIdx.x & 7u) - grp_) & 7)) * fchunk + r_ / jpf_; \
  if (fr_ >= nframes) return;                                                           \
  const int job = fr_ * jpf_ + r_ % jpf_
// three workgroups per CU for the stage form too: 48 KB of LDS at BAND_DEPTH 1, and an 80-register cap (4 dwords spill, in
// the candidate path); measured 1.07 -> 1.03 ms per 1024 x 1080p against two workgroups per CU at 89 registers
#if BAND_DEPTH == 1
#define STAGE_ATTR __attribute__((amdgpu_waves_per_eu(6, 6)))
#define STAGE_WGS 3
#else
#define STAGE_ATTR
#define STAGE_WGS 2
#endif
template <int MODE, int PRIO, int NCH, bool SPLIT = false>
__global__ __launch_bounds__(512) STAGE_ATTR void k_dense_band(BAND_ARGS)
{
  __shared__ __attribute__((aligned(16384))) uint8_t lds[BandLds<MODE, NCH>::bytes];
  BAND_JOB;
  dense_band_body<MODE, PRIO, NCH, SPLIT>(BAND_PASS);
}
// the compact-map form needs 42-47 KB of LDS: three workgroups per CU fit if the kernel stays within 80 VGPRs
// (6 waves per SIMD); measured 1.26 -> 1.18 ms per 1024 x 1080p and 1.44 -> 1.32 ms per 256 x 4K with the cap (6-8
// dwords spill)
template <int MODE, int PRIO, int NCH>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_dense_band_occ6(BAND_ARGS)
{
  __shared__ __attribute__((aligned(16384))) uint8_t lds[BandLds<MODE, NCH>::bytes];
  BAND_JOB;
  dense_band_body<MODE, PRIO, NCH>(BAND_PASS);
}

bool rcc_dense_band_supported(const rcc_handle* h, const uint8_t* d_grey, const uint8_t* d_bin)
{
  const rcc_config& c = h->cfg;
  return (c.width % 16 == 0) && (c.height % 4 == 0) && c.width >= 64 && c.height >= 8 &&
         ((long long)c.width * c.height < (1LL << 30)) && c.width < (1 << 16) &&
         ((reinterpret_cast<uintptr_t>(d_grey) & 15) == 0) && ((reinterpret_cast<uintptr_t>(d_bin) & 15) == 0);
}

int rcc_dense_allow_skip(const rcc_handle* h);

template <int MODE, int PRIO, int NCH, bool SPLIT = false>
static void launch_band(rcc_handle* h, const uint8_t* d_grey, int nframes, uint8_t* d_out, rcc_cand* d_cand, int32_t* d_cand_count,
                        int nbands, int nseg, int seg_tiles, int allow_skip, hipStream_t s)
{
  const rcc_config& c = h->cfg;
#ifdef RCC_EXPERIMENTS
  static const int fc_env = getenv("RCC_DENSE_FCHUNK") ? atoi(getenv("RCC_DENSE_FCHUNK")) : 0;
#else
  const int fc_env = 0;
#endif
  const int fchunk = fc_env > 0 ? fc_env : (nframes >= 1024 ? 16 : nframes >= 64 ? nframes / 64 : 1);   // frames per chunk (BAND_JOB)
  const long long njobs = (long long)nbands * nseg * ((nframes + 8 * fchunk - 1) / (8 * fchunk)) * 8 * fchunk;
  const int tp = rcc_flat_tp(c.height);
  if constexpr (SPLIT)      // (instantiated only by the two-kernel form: experiments library)
    hipLaunchKernelGGL((k_dense_band<MODE, PRIO, NCH, true>), dim3((unsigned)njobs), dim3(512), 0, s, d_grey, c.width, c.height, nbands, nseg, seg_tiles,
                       nframes, c.thr_min_contrast, c.harris_thresh, c.cand_margin, c.max_candidates, allow_skip, d_out, d_cand, d_cand_count, fchunk,
                       h->d_flat, tp);
  else if constexpr (MODE == 2)
    hipLaunchKernelGGL((k_dense_band_occ6<MODE, PRIO, NCH>), dim3((unsigned)njobs), dim3(512), 0, s, d_grey, c.width, c.height, nbands, nseg, seg_tiles,
                       nframes, c.thr_min_contrast, c.harris_thresh, c.cand_margin, c.max_candidates, allow_skip, d_out, d_cand, d_cand_count, fchunk,
                       (unsigned long long*)nullptr, 0);
  else
    hipLaunchKernelGGL((k_dense_band<MODE, PRIO, NCH>), dim3((unsigned)njobs), dim3(512), 0, s, d_grey, c.width, c.height, nbands, nseg, seg_tiles,
                       nframes, c.thr_min_contrast, c.harris_thresh, c.cand_margin, c.max_candidates, allow_skip, d_out, d_cand, d_cand_count, fchunk,
                       (unsigned long long*)nullptr, 0);
}

// h->want_thr (set by rcc_detect_batch): write the compact threshold map h->d_thr instead of the binary image
// split != 0: the two-kernel form -- this launch only thresholds and leaves the flat masks (h->d_flat), the corner
// stages follow in k_dense_runs (k_dense_runs.hip)
hipError_t rcc_launch_dense_band(rcc_handle* h, const uint8_t* d_grey, int nframes, uint8_t* d_bin,
                                 rcc_cand* d_cand, int32_t* d_cand_count, hipStream_t s, int split)
{
  const rcc_config& c = h->cfg;
  const int w = c.width, ht = c.height, th = ht >> 2;
  const int nbands = (w + BAND_W - 1) / BAND_W;
#ifdef RCC_EXPERIMENTS
  static const int memonly = getenv("RCC_DENSE_MEMONLY") ? atoi(getenv("RCC_DENSE_MEMONLY")) : 0;   // data movement only: WRONG results
#else
  const int memonly = 0;
#endif
  const bool narrow = (nbands == 1) && (w <= 2048);
  const bool thr = h->want_thr && h->d_thr && !memonly;
  // Segments per frame.  A job (one workgroup marching its segment) is a chain that takes as long alone as beside two
  // others on the CU, so the pass is rounds of jobs over the resident slots (3 workgroups per CU for the compact form, 2
  // for the stage form) as much as it is throughput: the count of jobs should fill whole rounds, and finer jobs pack
  // better when the active rows are bunched (scratch/t_dense_flat.py, 1024 x 1080p compact: 2 / 3 / 4 / 5 / 6 / 9 segments
  // 0.99 / 0.98 / 1.09 / 0.96 / 0.95 / 0.99 ms on the bench batch, 0.99 / 0.73 / 1.08 / 0.71 / 0.77 / 0.73 ms with the top half
  // of every frame flat).  Rule: up to 6 segments of at least 32 tile rows (16 while the launch does not fill the slots);
  // among them the largest count whose last round is within 2 % of the fullest.
  static const int cus = [] { hipDeviceProp_t p; int d = 0; return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&p, d) == hipSuccess) ? p.multiProcessorCount : 256; }();
  const long long slots = (long long)cus * (thr ? 3 : STAGE_WGS), base = (long long)nbands * nframes;
  int nseg = 1;
  {
    double best = 1e30;
    for (int n = 1; n <= 16; ++n) {
      const int st = (th + n - 1) / n;
      const bool fills = base * n >= slots;
      if (n > 1 && (st < 16 || (fills && (st < 32 || n > 6)))) break;
      const long long jobs = base * ((th + st - 1) / st);
      const double waste = (double)(((jobs + slots - 1) / slots) * slots) / (double)jobs;
      if (waste <= best * 1.02) { if (waste < best) best = waste; nseg = n; }
    }
  }
#ifdef RCC_EXPERIMENTS
  static const int nseg_env = getenv("RCC_DENSE_NSEG") ? atoi(getenv("RCC_DENSE_NSEG")) : 0;
#else
  const int nseg_env = 0;
#endif
  if (nseg_env > 0) nseg = nseg_env;
  const int seg_tiles = (th + nseg - 1) / nseg;
  nseg = (th + seg_tiles - 1) / seg_tiles;
  const int allow_skip = rcc_dense_allow_skip(h);
  h->bin_from_thr = thr ? 1 : 0;
  // split == 2: one independent wavefront per window (k_dense_wave.hip) -- the compact-map form only
  if (split == 2 && thr && rcc_dense_wave_supported(h, d_grey)) return rcc_launch_dense_wave(h, d_grey, nframes, d_cand, d_cand_count, s);
#ifdef RCC_EXPERIMENTS
  if (split == 1) {          // the two-kernel form (measurement only, librcc_hip_exp.so)
    const int tp = rcc_flat_tp(ht);
    const size_t need = rcc_flat_index(h->cfg.batch_capacity > nframes ? h->cfg.batch_capacity : nframes, 0, 0, nbands, tp) * sizeof(unsigned long long);
    if (need > h->flat_bytes) {
      if (h->d_flat) (void)hipFree(h->d_flat);
      h->d_flat = nullptr; h->flat_bytes = 0;
      hipError_t e = hipMalloc((void**)&h->d_flat, need);
      if (e != hipSuccess) return e;
      h->flat_bytes = need;
    }
    if (thr && narrow) { h->dense_kernel = "k_dense_band<2, 0, 8, true> + k_dense_runs"; launch_band<2, 0, 8, true>(h, d_grey, nframes, h->d_thr, d_cand, d_cand_count, nbands, nseg, seg_tiles, allow_skip, s); }
    else if (thr) { h->dense_kernel = "k_dense_band<2, 0, 9, true> + k_dense_runs"; launch_band<2, 0, 9, true>(h, d_grey, nframes, h->d_thr, d_cand, d_cand_count, nbands, nseg, seg_tiles, allow_skip, s); }
    else if (narrow) { h->dense_kernel = "k_dense_band<0, 0, 8, true> + k_dense_runs"; launch_band<0, 0, 8, true>(h, d_grey, nframes, d_bin, d_cand, d_cand_count, nbands, nseg, seg_tiles, allow_skip, s); }
    else { h->dense_kernel = "k_dense_band<0, 0, 9, true> + k_dense_runs"; launch_band<0, 0, 9, true>(h, d_grey, nframes, d_bin, d_cand, d_cand_count, nbands, nseg, seg_tiles, allow_skip, s); }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return rcc_launch_dense_runs(h, d_grey, nframes, h->d_flat, tp, d_cand, d_cand_count, s);
  }
  if (memonly) { h->dense_kernel = "k_dense_band<1, 0, 9, false>"; launch_band<1, 0, 9>(h, d_grey, nframes, d_bin, d_cand, d_cand_count, nbands, nseg, seg_tiles, allow_skip, s); return hipGetLastError(); }
#endif
  if (thr && narrow) { h->dense_kernel = "k_dense_band_occ6<2, 1, 8>"; launch_band<2, 1, 8>(h, d_grey, nframes, h->d_thr, d_cand, d_cand_count, nbands, nseg, seg_tiles, allow_skip, s); }
  else if (thr) { h->dense_kernel = "k_dense_band_occ6<2, 1, 9>"; launch_band<2, 1, 9>(h, d_grey, nframes, h->d_thr, d_cand, d_cand_count, nbands, nseg, seg_tiles, allow_skip, s); }
  else if (narrow) { h->dense_kernel = "k_dense_band<0, 1, 8, false>"; launch_band<0, 1, 8>(h, d_grey, nframes, d_bin, d_cand, d_cand_count, nbands, nseg, seg_tiles, allow_skip, s); }
  else { h->dense_kernel = "k_dense_band<0, 1, 9, false>"; launch_band<0, 1, 9>(h, d_grey, nframes, d_bin, d_cand, d_cand_count, nbands, nseg, seg_tiles, allow_skip, s); }
  return hipGetLastError();
}
