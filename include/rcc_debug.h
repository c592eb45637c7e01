/*
 * rcc_debug.h -- test taps, HIP-event timers and A/B switches of librcc_hip.so.  NOT part of the drop-in boundary
 * (include/rcc.h): nothing here stands in for an interface of the reference, and a host built on rcc.h never needs it.
 *
 * Why the symbols are still in the shipped library: the parity tests (tests/) compare the HIP path with the oracle
 * STAGE BY STAGE through the C ABI, so they need the intermediate images and lists (rcc_debug_fetch_*), and bench.py
 * needs per-kernel HIP-event times for its roofline leg (rcc_time_*).  Every entry point here is read-only with
 * respect to results: taps copy buffers out, timers launch a pass repeatedly, and each switch selects between kernel
 * variants that are tested bit-identical (tests/test_gpu_parity.py::test_kernel_variants_bit_identical).
 *
 * What is NOT in the shipped library: anything that can change a result or exists only for a one-off experiment --
 * environment-variable knobs (RCC_DENSE_MEMONLY runs a pass's data movement without its arithmetic: wrong results;
 * RCC_DENSE_NSEG, RCC_DENSE_FCHUNK, RCC_RUNS_NSEG, RCC_INGEST_FPB, RCC_PNP_SOLVER) and rcc_debug_overlap.  Those are
 * compiled only with -DRCC_EXPERIMENTS (make -C robot_camera_calibration_amd/csrc EXPERIMENTS=1 builds
 * librcc_hip_exp.so beside the product library); the default build reads no environment variable at all.
 */
#ifndef RCC_DEBUG_H_
#define RCC_DEBUG_H_

#include "rcc.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- A/B switches between bit-identical variants (each returns the previous setting) -------------------------------- */
/* threshold + corner pass: 0 = generic LDS tiles (any geometry), 1 = band kernel (k_dense_band.hip), 2 = strip march
 * (k_dense_fast.hip), 3 = band sweep + corner kernel on the active rows (k_dense_runs.hip: experiments library only -- the
 * product library runs 1 in its place), 4 = one independent wavefront
 * per window where the binary image is kept as the compact map, the band kernel otherwise (k_dense_wave.hip);
 * -1 = automatic (4 where the geometry allows it, else 2, else 0) */
int rcc_set_dense_variant(rcc_handle* h, int variant);
/* undistort + grey pass: 0 = gather (any geometry), 1 = LDS-staged 128 x 16 destination tiles with the tabulated map, 2 = staged,
 * map recomputed per block (the form a handle falls back to when the table cannot be allocated), 3 = staged with the 128 x 8 tiles
 * of rounds 1-3 (experiments library only: the product library runs 1 in its place); -1 = automatic */
int rcc_set_ingest_variant(rcc_handle* h, int variant);
/* 1 (default): the marching dense kernels skip the corner stages on wave-rows whose tiles are all flat (exact); 0: never */
int rcc_set_dense_skip(rcc_handle* h, int on);
/* 1 (default): the checkerboard path runs lattice indexing and the pose solve of a frame in one kernel; 0: two kernels */
int rcc_set_fuse_grid_pnp(rcc_handle* h, int on);
/* n > 1: cut each batch into n chunks (>= 64 frames) alternating over two internal streams; default 1 (a single pass is
 * faster at every size measured on MI355X: DESIGN.md section 5).  Per-stage timings exist only for n <= 1. */
int rcc_set_pipeline(rcc_handle* h, int nchunks);
/* host-resident batches (RCC_MEM_HOST) go over as a pipeline of chunks, each chunk's kernels under the next chunks' copies:
 * frames per chunk (0 = automatic, about 192 MiB; < 0 = one copy of the whole batch, then the kernels -- the A/B form).
 * Same records either way.  Returns the previous setting. */
int rcc_set_host_chunk(rcc_handle* h, int frames_per_chunk);
/* tag scenes: width of k_subpix's grid -- a wave walks its frame's candidate list with this stride (0 = automatic: at least 64,
 * wider for small batches; k_subpix.hip).  Same refined positions at any width.  Returns the previous setting. */
int rcc_set_subpix_grid(rcc_handle* h, int width);
/* PnP mapping: 0 = one lane per target, 1 = one wavefront per target with more than 8 points, -1 = automatic (= 1) */
int rcc_set_pnp_variant(rcc_handle* h, int variant);

/* ---- timers (HIP events on the launch stream) ------------------------------------------------------------------------ */
/* stage times of the last synchronous rcc_detect_batch / stage call, ms: [0] ingest, [1] threshold+corner,
 * [2] list + sub-pixel (+ grid when not fused), [3] pose (+ grid when fused), [4] d2h (-1 where not recorded: a chunked
 * batch has no separate stage durations).  After rcc_detect_batch_collect: the stages as they ran inside that streamed
 * step.  Returns the slots written. */
int rcc_last_timings(const rcc_handle* h, float* ms, int32_t n);
/* the batch rcc_detect_batch_collect just handed back, ms: [0] device time from the moment its stream reached the batch
 * to its records lying in pinned host memory, [1] device idle time in front of it (end of the previous submission ->
 * begin of this one; ~0 while the host keeps a batch ahead, > 0 where the host submitted late; -1 for the first
 * submission), [2..6] = the five stage times of rcc_last_timings.  Returns the slots written. */
int rcc_last_step_times(const rcc_handle* h, float* ms, int32_t n);
/* the engine clock this device holds under a vector-issue load and the cost of one vector wave-instruction, measured by
 * this call (k_probe.hip): a launch of about ms_target milliseconds of the threshold + corner pass's instruction classes
 * (packed 16-bit, dot2, perm, DPP, three-operand adds) at waves_per_simd waves per SIMD on every CU.
 * out6: [0] clock in MHz (delta s_memtime / delta s_memrealtime x 100 MHz, median over the waves), [1] ns per
 * wave-instruction per SIMD (HIP events around the launch), [2] = [1] x [0] cycles, [3] ms of the launch, [4] / [5]
 * lowest / highest clock any wave saw. */
int rcc_debug_measure_clock(rcc_handle* h, int32_t waves_per_simd, float ms_target, double* out6);
/* name(s) of the kernel(s) the last threshold + corner launch used, as rocprofv3's kernel trace prints them */
const char* rcc_last_dense_kernel(const rcc_handle* h);
/* mean ms of `reps` back-to-back launches of the threshold + corner pass.  d_bin == NULL: the form rcc_detect_batch
 * launches (binary image left as the compact threshold map in the handle; nframes <= batch_capacity) */
int rcc_time_dense(rcc_handle* h, const void* d_grey, int32_t nframes, void* d_bin, void* d_cand,
                   void* d_cand_count, int32_t reps, float* mean_ms);
int rcc_time_ingest(rcc_handle* h, const void* d_frames, int32_t nframes, void* d_grey,
                    int32_t reps, float* mean_ms);
/* mean ms of `reps` plain streaming copies (16 B per lane, four loads in flight) of nbytes between two device buffers
 * (16-byte aligned, nbytes a multiple of 16): the yardstick bench.py quotes beside the bandwidth-bound passes */
int rcc_time_copy(rcc_handle* h, const void* d_src, void* d_dst, int64_t nbytes, int32_t reps, float* mean_ms);
/* counter calibration for the PMC passes: one dword-per-lane copy and one 16 B-per-lane copy of a known byte count */
int rcc_debug_calib_copy(rcc_handle* h, const void* d_src, void* d_dst, int64_t nbytes);

/* ---- test taps (host pointers; any may be NULL) ---------------------------------------------------------------------- */
/* intermediate lists / images of the handle's last rcc_detect_batch.  pre / kept: nframes x 256 (2048 for fiducials)
 * entries {int16 x, int16 y, int32 score}; pre_xy / kept_xy: the same x 2 doubles; cand: nframes x max_candidates.
 * `bin` is expanded from the compact threshold map on demand. */
int rcc_debug_fetch_lists(rcc_handle* h, int32_t nframes, void* pre, int32_t* npre, double* pre_xy,
                          void* kept, double* kept_xy);
int rcc_debug_fetch_images(rcc_handle* h, int32_t nframes, void* grey, void* bin, void* cand,
                           int32_t* cand_count);
/* intermediates of one PnP solve: out[59] = H[9], initial pose[6], JtJ[36], Jte[6], |e|^2, status * 10 + ok */
int rcc_debug_pnp_probe(rcc_handle* h, const double* obj, const double* img, int32_t n, const double* K,
                        const double* D, int32_t dist_model, double* out);

#ifdef RCC_EXPERIMENTS
/* measurement-only forms of the threshold + corner pass, bit-identical to the product's (tests/test_experiments_library.py runs
 * them against librcc_hip_exp.so): dense variant 3 -- the band sweep + a corner kernel on the active rows
 * (k_dense_runs.hip; the product library runs variant 1 in its place) -- and k_dense_wave as gangs of eight windows per
 * workgroup that meet at a barrier every sync_rows tile rows (a power of two; 0 = every window on its own); segments per
 * frame for that form (0 = default).  Returns the previous sync_rows. */
int rcc_set_dense_gang(rcc_handle* h, int sync_rows, int segments);
/* experiment (scratch/t_overlap.py): the ingest pass and the threshold+corner pass over independent buffers, back to
 * back on one stream (mode 0) or launched together on two streams (mode 1); mean milliseconds per pair. */
int rcc_debug_overlap(rcc_handle* h, const void* d_frames, int32_t nframes, void* d_grey_out, const void* d_grey_in,
                      void* d_cand, void* d_cand_count, int32_t mode, int32_t reps, float* mean_ms);
/* experiment: the wave-per-window threshold + corner pass reads the grey rows of frame (f mod m) -- a working set small enough
 * to stay in the Infinity Cache; m = 0: off.  Results are then those of the wrong frames. */
int rcc_set_dense_fmod(rcc_handle* h, int32_t m);
/* experiment (round 4): mode 1 -- a streamed board batch's lattice + pose kernel runs on a stream of the handle's own, under the next
 * batch's ingest pass; 0 off.  Device-resident batches without corner tables and
 * without record tables only.  Result: DESIGN.md section 5, "Measured, not kept". */
int rcc_set_tail_overlap(rcc_handle* h, int32_t mode);
/* experiment (scratch/t_grid_trace.py): per-frame phase time stamps of k_grid_pnp -- d_buf: nframes x 24 int64 (device), slots
 * 0 start, 1 seeds done, 3 growth starts, 2 growth done, 5 lattice done, 6 pose done (10-ns ticks), 7 = seed attempt * 1000 + labels,
 * 8 homography starts, 9 DLT done, 10 refinement done, 11 homography returned, 12 initial pose done, 13 = refinement iterations,
 * 16..23 = time spent in: H accumulate (full), H accumulate (trial), H 8x8 solve, H rest, pose accumulate (full), pose accumulate (trial),
 * pose 6x6 solve, (unused);
 * NULL switches it off. */
int rcc_debug_grid_trace(rcc_handle* h, void* d_buf);
#endif

#ifdef __cplusplus
}
#endif
#endif /* RCC_DEBUG_H_ */
