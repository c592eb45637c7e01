// k_mix.hip -- EXPERIMENT (librcc_hip_exp.so only; rcc_debug_overlap mode 2, scratch/t_overlap.py): the undistort + grey pass of
// one set of frames and the threshold + corner pass of another, side by side in ONE launch.
//
// The idea: the two passes of a batch lean on different resources -- the ingest pass on HBM (5.2 TB/s of a 3 : 1 read : write
// mix, vector ALU ~30 % busy), the threshold + corner pass on vector-instruction issue (85 % of the issue rate) -- so run
// together each should fill what the other leaves idle.  Two streams do not mix them (the dispatcher hands a second queue's
// workgroups out only once a chip-filling grid has been dispatched: DESIGN.md section 5, "Measured, not kept"), so here the
// mixing is done by hand: one grid whose workgroups take one of two roles, interleaved in dispatch order in the ratio of
// the two job counts, so that every CU holds both kinds at any time:
//   * ingest role: one 128 x 8 destination tile over `fpb` frames, exactly k_ingest_staged's workgroup (ingest_staged.h);
//   * dense role: four independent single-wavefront window jobs, exactly k_dense_wave's (dense_wave_body.h), each with
//     its own 4-KiB ring -- no barrier, so the role's four waves retire one by one.
// Both roles keep their XCD-aware deals: the workgroup's XCD is blockIdx.x & 7 as before, and what the stand-alone kernels
// derive from blockIdx.x >> 3 they get here from the role's own running index on that XCD.  Outputs are bit-identical to
// the two stand-alone kernels (same device functions, same job decomposition).
//
// MEASURED (1024 + 1024 x 1080p, two boxes): 2.56-2.66 ms for the mixed launch against 2.52-2.59 ms for the two kernels back to
// back (1.74 + 0.77 alone) -- no gain, as in round 1 with the kernels of that time.  With the dense role's input made
// resident in the Infinity Cache (rcc_set_dense_fmod 48) the mixed launch takes 2.54-2.56 ms: HBM is not what the two share.
// What they share is wave slots: both passes deliver in proportion to the wavefronts a CU holds (the ingest pass needs its
// 32 waves per CU to keep ~70 KB of loads in flight, the window jobs are single dependent chains of which a SIMD needs six
// to fill its issue slots), and 74 registers allow 24 waves per CU for the two together.  A mix would need an ingest role
// whose bytes in flight cost no registers (source rows by LDS-DMA), i.e. another ingest kernel.  Not in the product library.
#include "ingest_staged.h"
#include "dense_wave_body.h"

#define MIX_LDS (4 * WAVE_RING * 1024)
static_assert(MIX_LDS >= ST_TILE_LDS + 16, "the ingest tile must fit the workgroup's LDS");

template <int NCH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 6)))
void k_mix(// ingest role
           const uint8_t* __restrict__ frames, int64_t frame_bytes, int stride, int w, int h, rcc_cam cam, uint8_t* __restrict__ grey_out,
           int n_in, int fpb, int ntx, int ntiles, int per_xcd, const int2* __restrict__ map, const int4* __restrict__ tilebox,
           // dense role
           const uint8_t* __restrict__ grey_in, int nbands, int nwin, int nseg, int seg_tiles, int n_dn, int min_contrast, int hthresh,
           int margin, int cap, int allow_skip, uint8_t* __restrict__ thr_map, rcc_cand* __restrict__ cand, int32_t* __restrict__ cand_count, int fchunk,
           // the interleave: per XCD, ki ingest workgroups and kd dense workgroups (of four window jobs)
           int ki, int kd, int dense_fmod /* experiments: dense_wave_body.h */)
{
  __shared__ __attribute__((aligned(1024))) uint8_t lds[MIX_LDS];
  const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
  // slot k of the XCD's ki + kd is a dense slot where floor((k + 1) kd / (ki + kd)) steps: the dense workgroups are spread
  // evenly through the dispatch order, d0 of them before slot k
  const long long tot = (long long)ki + kd;
  const int d0 = (int)(((long long)k * kd) / tot), d1 = (int)(((long long)(k + 1) * kd) / tot);
  if (d1 != d0) {
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned vblock = ((unsigned)(4 * d0 + wib) << 3) | (unsigned)xcd;
    dense_wave_body<WAVE_PRIO, 1>(grey_in, w, h, nbands, nwin, nseg, seg_tiles, n_dn, min_contrast, hthresh, margin, cap, allow_skip, thr_map, cand, cand_count,
                                  fchunk, dense_fmod, vblock, 0, lds + wib * (WAVE_RING * 1024));
    return;
  }
  const int kk = k - d0;
  const int bz = kk / per_xcd;
  const int tile = xcd * per_xcd + (kk - bz * per_xcd);
  if (tile >= ntiles) return;                       // block-uniform
  int* s_flag = reinterpret_cast<int*>(lds + ST_TILE_LDS);
  ingest_staged_body<NCH>(frames, frame_bytes, stride, w, h, cam, grey_out, n_in, fpb, ntx, tile, bz, threadIdx.x, lds, s_flag, 0, 1, map, tilebox);
}

int rcc_dense_allow_skip(const rcc_handle* h);

hipError_t rcc_launch_mix(rcc_handle* h, const uint8_t* d_frames, int n_in, uint8_t* d_grey_out,
                          const uint8_t* d_grey_in, int n_dn, uint8_t* d_thr, rcc_cand* d_cand, int32_t* d_cand_count, hipStream_t s, bool* done)
{
  const rcc_config& c = h->cfg;
  *done = false;
  if (n_in <= 0 && n_dn <= 0) { *done = true; return hipSuccess; }
  rcc_ingest_plan ip = {};
  int ki = 0;
  if (n_in > 0) {
    bool staged = false;
    const int tile8 = h->ingest_tile8;
    h->ingest_tile8 = 1;                 // this launch's ingest role runs 128 x 8 tiles in 256-thread workgroups
    hipError_t e = rcc_ingest_staged_plan(h, d_frames, n_in, s, &ip, &staged);
    h->ingest_tile8 = tile8;
    if (e != hipSuccess) return e;
    if (!staged) return hipSuccess;
    ki = ip.per_xcd * ip.ngroups;
  }
  rcc_wave_plan wp = {};
  int kd = 0;
  if (n_dn > 0) {
    if (!d_thr || !rcc_dense_wave_supported(h, d_grey_in)) return hipSuccess;
    rcc_dense_wave_plan(h, n_dn, &wp);
    kd = (int)((wp.njobs / 8 + 3) / 4);            // njobs is a multiple of 8: the same count for every XCD
  }
  const long long blocks = 8LL * ((long long)ki + kd);
  if (blocks <= 0 || blocks > 0x7FFFFFFFLL) return hipSuccess;
  h->bin_from_thr = 1;
#define MIX_ARGS d_frames, c.frame_bytes, c.stride_bytes, c.width, c.height, ip.cam, d_grey_out, n_in, ip.fpb, ip.ntx, ip.tiles, ip.per_xcd,                 \
                 (const int2*)ip.map, (const int4*)ip.tilebox, d_grey_in, wp.nbands, wp.nwin, wp.nseg, wp.seg_tiles, n_dn, c.thr_min_contrast, c.harris_thresh, \
                 c.cand_margin, c.max_candidates, rcc_dense_allow_skip(h), d_thr, d_cand, d_cand_count, wp.fchunk, ki, kd, h->dense_fmod
  if (c.pixfmt == RCC_PIX_BGR8) hipLaunchKernelGGL((k_mix<3>), dim3((unsigned)blocks), dim3(256), 0, s, MIX_ARGS);
  else hipLaunchKernelGGL((k_mix<1>), dim3((unsigned)blocks), dim3(256), 0, s, MIX_ARGS);
#undef MIX_ARGS
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) *done = true;
  return e;
}
