// rcc_internal.h -- shared declarations of the HIP implementation behind include/rcc.h.
// gfx950 (MI355X) only; wavefront = 64.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rcc.h"

#define RCC_WAVE 64
#define RCC_MAX_KEPT 256
#define RCC_THR_PITCH 512     // bytes per tile row of one band in the compact threshold map (480 used at 1920 columns)
#define RCC_BAND_W 1920        // output columns per band of the band kernel
#define RCC_HOST_CHUNKS 64      // at most this many chunks per host-resident batch
#define RCC_SUBT_RING 4          // submissions whose timing events are kept (rcc_detect_batch_submit)
#define RCC_PROBE_INSTS_PER_ITER 64   // k_probe.hip: vector instructions per loop iteration of the issue probe

// Flat masks of the two-kernel threshold + corner pass (k_dense_band.hip sweep -> k_dense_runs.hip): one 64-bit word per
// (frame, band, window of the band, tile row): bit l = lane l's 4x4 tile has a dilated contrast below min_contrast.
// A window's words are contiguous: entry x + 1 describes tile row x, x = -1 .. th (tp >= th + 2 entries).
__host__ __device__ inline size_t rcc_flat_index(int f, int band, int window, int nbands, int tp)
{
  return (((size_t)f * nbands + band) * 8 + window) * (size_t)tp;
}
inline int rcc_flat_tp(int height) { return (((height >> 2) + 2) + 7) & ~7; }

struct rcc_cand {  // dense-pass list entry, 8 bytes
  int16_t x, y;
  int32_t score;
};

// camera parameters passed by value to kernels
struct rcc_cam {
  double fx, fy, cx, cy;
  double D[8];
  int model;
  int solver;   // PnP normal-equation solver: 0 eigen (as published), 1 Cholesky + eigen fallback
};

struct rcc_subpix_params {
  int win, max_iter;
  double eps2;
  double m1[15];  // exp(-(k/win)^2), k=-win..win, computed on the host (libm)
};

// k_subpix's per-lane sample table (the window geometry is fixed per handle): built on the host at rcc_create
#define RCC_SP_PT 5     // ceil(17 * 17 / 64): patch samples per lane at the largest window
#define RCC_SP_GT 4     // ceil(15 * 15 / 64): gradient samples per lane
struct rcc_subpix_lane {
  int32_t poff[RCC_SP_PT];   // (i - win - 1) * w + (j - win - 1) of patch sample lane + 64 t
  int32_t goff[RCC_SP_GT];   // (i + 1) * pw + (j + 1): centre of gradient sample lane + 64 t in the patch
  int8_t gpx[RCC_SP_GT], gpy[RCC_SP_GT];   // j - win, i - win
  int32_t pad;
  double gm[RCC_SP_GT];      // m1[i] * m1[j], 0 beyond the window
};

struct rcc_handle {
  rcc_config cfg;
  int device;
  hipStream_t stream;
  int undist;  // cfg.undistort && model != NONE
  // scratch sized for batch_capacity frames
  uint8_t* d_grey;
  uint8_t* d_bin;
  uint8_t* d_thr;           // compact threshold map of rcc_detect_batch: per frame [band][tile row][RCC_THR_PITCH] bytes,
                            // one per 4x4 tile: 255 = flat tile (binary value 127), else the level (pixel > level ? 255 : 0)
  unsigned long long* d_flat;   // flat masks of the split threshold + corner pass (rcc_flat_index), allocated on first use
  size_t flat_bytes;
  double* rec_table[2];     // rcc_set_record_tables: the caller's device tables, one per result slot (NULL: none)
  int rec_offset;           // global index of the batch's first frame in those tables
  int rec_capacity;         // slots each of those tables holds (the packer zeroes the slots a shorter batch leaves)
  const char* dense_kernel; // name(s) of the kernel(s) the last threshold + corner launch used, as rocprofv3 prints them
  void* d_map;              // staged ingest: Q5 map of the handle's camera (w * h int2) and the tiles' source boxes (int4 each),
  void* d_tilebox;          //   tabulated at the first staged launch (k_ingest_map)
  int map_failed;           // the tables could not be allocated: the kernel recomputes the map per block
  int map_th;               // tile height the source boxes in d_tilebox were tabulated for (8 or 16)
  int ingest_tile8;         // rcc_set_ingest_variant(3): staged with 128 x 8 tiles even where 128 x 16 fit (A/B, tests)
  int ingest_table;         // 1 (default): use the tables; 0: recompute (A/B, tests)
  int tail_overlap;         // experiments only (rcc_set_tail_overlap): 1 = the lattice + pose kernel of a streamed batch on its own stream, under the next batch's ingest pass
  hipStream_t tail_stream;
  hipEvent_t tail_val[2], tail_mark[2], tail_done[2];
  int tail_pending;         // 1 + slot whose tail may still be running on tail_stream
  int fuse_grid_pnp;        // 1 (default): board validation / indexing and pose in one kernel (checkerboard, wave-per-board solver)
  int keep_bin;             // rcc_set_keep_binary: rcc_detect_batch writes the full binary image (default 0: the compact map)
  int want_thr;             // set by rcc_detect_batch: the dense pass may write d_thr instead of the full binary image
  int bin_from_thr;         // set by the dense launcher: this batch's binary image exists only as d_thr
  rcc_cand* d_cand;
  int32_t* d_cand_count;
  rcc_cand* d_pre;        // B x 256
  int32_t* d_npre;        // B
  double* d_pre_xy;       // B x 256 x 2
  double* d_ref_xy;       // fiducial handles: B x kept_cap x 2, refined corner of list entry i where it is a corner of a decoded quad (refine_edges form)
  rcc_cand* d_kept;       // B x 256 (validated, rounded refined pixel)
  double* d_kept_xy;      // B x 256 x 2
  rcc_frame_corners* d_fc;  // B
  rcc_detection* d_det;     // B x max_targets
  int32_t* d_ndet;          // B
  uint8_t* d_stage;         // staging for host-resident input frames
  size_t stage_bytes;
  // pnp batch scratch (grown on demand)
  double* d_pnp_buf;
  size_t pnp_buf_bytes;
  rcc_detection* h_det;     // pinned
  int32_t* h_ndet;          // pinned
  // rcc_detect_batch_submit / _collect: a second pinned result slot, so that the host can unpack batch k while the
  // device runs batch k+1 (slot 0 = h_det / h_ndet above)
  rcc_detection* h_det2;
  int32_t* h_ndet2;
  hipEvent_t sub_ev[2];     // results of the submission in slot i are in pinned memory
  // the per-frame corner tables of a submission (rcc_frame_corners, 6 KB per frame) go to the host on a copy stream, under the
  // next batch's ingest and threshold + corner passes: fc_ready[i] = the tables of slot i are complete (on the batch's stream),
  // fc_done[i] = they are in the caller's memory (on the copy stream); the next batch's list stage, the first kernel that writes
  // d_fc again, waits for fc_done
  hipEvent_t fc_ready[2], fc_done[2];
  hipStream_t fc_stream;    // the copy stream of those tables (its own: the chunk copies of a host-resident batch use pstream[])
  int fc_pending;           // slot + 1 of a corner-table copy the next launch_targets must wait for; 0 none
  unsigned char sub_has_fc[2];
  rcc_frame_corners* h_fc[2];       // pinned landing areas of those copies (allocated with the first submission that asks for corner tables):
  rcc_frame_corners* sub_fc_dst[2]; //   the caller's array may be pageable, and a device-to-pageable copy would hold the submitting thread
                                    //   until the batch is done; rcc_detect_batch_collect copies slot -> caller
  // timing events of the streaming form, a ring over the last RCC_SUBT_RING submissions (submission n uses entry n mod ring, so the
  // previous submission's end is still readable when this one is collected): [0] the stream reaches the batch, [1] ingest done,
  // [2] threshold + corner pass done, [3] list + sub-pixel (+ quads) done, [4] pose done, [5] records in pinned host memory
  hipEvent_t sub_t_ev[RCC_SUBT_RING][6];
  unsigned sub_t_seq[RCC_SUBT_RING];     // submission number + 1 whose events the entry holds (0: none)
  hipStream_t sub_t_stream[RCC_SUBT_RING];
  unsigned char sub_t_staged[RCC_SUBT_RING];   // 1: stage events [1..4] were recorded (0: chunked host-input pipeline, only [0] and [5])
  float last_step_ms[7];    // after rcc_detect_batch_collect: [0] device time of that batch (events 0 -> 5), [1] device idle in front of it
                            // (previous submission's [5] -> this [0]; -1 unknown), [2..6] ingest, threshold + corner, list + sub-pixel, pose, d2h
  int sub_nframes[2];       // frames of the submission in slot i (0: slot free)
  hipStream_t sub_stream[2];
  unsigned sub_head, sub_tail;   // submissions issued / collected
  double* d_board_obj;      // 256 x 3 object points of the board
  double* d_img_scratch;    // B x 256 x 2 image points handed to the solver
  hipEvent_t ev[8];
  float last_ms[5];
  hipStream_t pstream[2];   // chunk streams of rcc_detect_batch's pipeline
  hipEvent_t pev[3];        // [0] input ready on the caller's stream, [1..2] chunk streams drained
  hipEvent_t cev[RCC_HOST_CHUNKS];   // host-input pipeline: chunk c's copy has landed in the staging buffer (created on first use)
  int host_chunk_frames;    // 0: automatic (about 192 MiB per chunk); > 0: frames per chunk; < 0: one copy of the whole batch, then the kernels
  int pipeline_chunks;      // 0/1: one pass over the whole batch on one stream; n > 1: n chunks alternating over two streams
  int dense_variant, ingest_variant;
  int dense_fmod;           // experiments only (rcc_set_dense_fmod): the pass reads the grey rows of frame f mod this; 0 off
  int dense_gang_sync;      // k_dense_wave: 0 = every window its own workgroup; n (a power of two) = gangs of eight windows meeting every n tile rows
  int dense_gang_seg;       // segments per frame of the gang form (0: as the single-window form)
  int dense_skip;           // 1: the fast dense kernel may skip flat wave-rows (exact); 0: never (A/B, tests)
  int pnp_variant;          // -1 auto, 0 lane per target, 1 wavefront per target (board)
  int pnp_solver;           // 0 eigen, 1 Cholesky (default)
  int pnp_use_mfma;         // cfg.pnp_use_mfma / rcc_set_pnp_mfma: 4-point tag poses accumulate J^T J, J^T e on the matrix cores
  int kept_cap;             // stride of the per-frame suppressed-list buffers: 256 (board) or 2048 (fiducials)
  uint64_t* d_family;       // fiducial family table (device copy)
  int pnp_wave_hint;        // set per rcc_solve_pnp_batch call: max points per target > 8
  rcc_subpix_params sp;
  int subpix_grid;           // rcc_set_subpix_grid: width of k_subpix's grid in tag scenes; 0 = automatic
  rcc_subpix_lane* d_sp_tab;   // 64 entries
  void* d_synth_tmp;         // synthetic camera with optics: the integer images between its passes (grown on demand)
  size_t synth_tmp_bytes;
  char err[256];
};

// ---- launchers (each returns hipError_t of the launch) ---------------------------------------
hipError_t rcc_launch_ingest(rcc_handle* h, const uint8_t* d_frames, int nframes, uint8_t* d_grey, hipStream_t s);
// launch geometry of the staged undistort + grey pass (k_ingest.hip) and of the wave-per-window threshold + corner pass
// (k_dense_wave.hip), worked out by the files that own the kernels and shared with the launch that runs both side by side (k_mix.hip)
struct rcc_ingest_plan {
  rcc_cam cam;
  int fpb, ntx, tiles, per_xcd, ngroups;     // frames per workgroup, tiles per row, tiles, tiles per XCD, frame groups
  int th;                                    // tile height: 8 or 16 destination rows (128 columns)
  const void* map; const void* tilebox;      // the tabulated Q5 map and the tiles' source boxes (null: recomputed per workgroup)
};
struct rcc_wave_plan {
  int nbands, nwin, nseg, seg_tiles, fchunk;
  long long njobs;                            // single-wavefront jobs, padded to whole deals (dense_wave_body.h)
};
// *staged = false: this configuration does not take the staged form (nothing else is filled in)
hipError_t rcc_ingest_staged_plan(rcc_handle* h, const uint8_t* d_frames, int nframes, hipStream_t s, rcc_ingest_plan* p, bool* staged);
void rcc_dense_wave_plan(const rcc_handle* h, int nframes, rcc_wave_plan* p);
// ingest of frames [0, n_in) of d_frames -> d_grey_out beside the threshold + corner pass of the n_dn frames at d_grey_in (an earlier
// chunk of the batch), one launch; either count may be 0.  *done = false: not applicable here, nothing was launched
hipError_t rcc_launch_mix(rcc_handle* h, const uint8_t* d_frames, int n_in, uint8_t* d_grey_out,
                          const uint8_t* d_grey_in, int n_dn, uint8_t* d_thr, rcc_cand* d_cand, int32_t* d_cand_count, hipStream_t s, bool* done);
bool rcc_dense_wave_supported(const rcc_handle* h, const uint8_t* d_grey);
hipError_t rcc_launch_dense_wave(rcc_handle* h, const uint8_t* d_grey, int nframes, rcc_cand* d_cand, int32_t* d_cand_count, hipStream_t s);
hipError_t rcc_launch_dense(rcc_handle* h, const uint8_t* d_grey, int nframes, uint8_t* d_bin,
                            rcc_cand* d_cand, int32_t* d_cand_count, hipStream_t s);
bool rcc_dense_band_supported(const rcc_handle* h, const uint8_t* d_grey, const uint8_t* d_bin);
hipError_t rcc_launch_dense_runs(rcc_handle* h, const uint8_t* d_grey, int nframes, const unsigned long long* d_flat, int flat_tp,
                                 rcc_cand* d_cand, int32_t* d_cand_count, hipStream_t s);
hipError_t rcc_launch_pack_records(rcc_handle* h, int nframes, int frame_offset, double* d_table, hipStream_t s);
hipError_t rcc_launch_list(rcc_handle* h, const rcc_cand* d_cand, const int32_t* d_cand_count,
                           int nframes, hipStream_t s);
hipError_t rcc_launch_subpix(rcc_handle* h, const uint8_t* d_grey, int nframes, hipStream_t s);
hipError_t rcc_launch_validate(rcc_handle* h, const uint8_t* d_grey, const uint8_t* d_bin, int nframes, hipStream_t s);
hipError_t rcc_launch_grid(rcc_handle* h, const uint8_t* d_grey, const uint8_t* d_bin, int nframes, hipStream_t s);
hipError_t rcc_launch_expand_bin(rcc_handle* h, const uint8_t* d_grey, int nframes, uint8_t* d_bin, hipStream_t s);
hipError_t rcc_launch_fid(rcc_handle* h, const uint8_t* d_grey, int nframes, hipStream_t s);
hipError_t rcc_launch_pnp_tags(rcc_handle* h, int nframes, hipStream_t s);
hipError_t rcc_launch_pnp_board(rcc_handle* h, int nframes, hipStream_t s);
hipError_t rcc_launch_pnp_tags_mfma(rcc_handle* h, int nframes, rcc_cam cam, hipStream_t s);
hipError_t rcc_launch_grid_pnp(rcc_handle* h, const uint8_t* d_grey, const uint8_t* d_bin, int nframes, hipStream_t s);
hipError_t rcc_launch_grid_pnp_only(rcc_handle* h, int nframes, hipStream_t s, int lean);
#ifdef RCC_EXPERIMENTS
hipError_t rcc_launch_marker(hipStream_t s);
#endif
bool rcc_grid_pnp_applicable(const rcc_handle* h);
hipError_t rcc_launch_pnp_generic(rcc_handle* h, const double* d_obj, const double* d_img,
                                  const int32_t* d_off, const int32_t* d_npts, int ntargets,
                                  rcc_cam cam, double* d_rvec, double* d_tvec, double* d_rms,
                                  int32_t* d_status, int32_t* d_iters, hipStream_t s);
hipError_t rcc_launch_pnp_probe(const double* d_obj, const double* d_img, int n, rcc_cam cam, double* d_out, hipStream_t s);
hipError_t rcc_launch_calib_copy(const void* src, void* dst, size_t nbytes, hipStream_t s);
hipError_t rcc_launch_copy_x4(const void* src, void* dst, size_t nbytes, hipStream_t s);
hipError_t rcc_launch_issue_probe(int blocks, int iters, unsigned long long* d_stamps, unsigned* d_sink, hipStream_t s);
hipError_t rcc_launch_rodrigues(int dir, const double* d_in, int n, double* d_out, hipStream_t s);
hipError_t rcc_launch_synth(rcc_handle* h, const rcc_synth_params* sp, const double* d_poses,
                            int nframes, int first_index, uint8_t* d_frames, hipStream_t s);
