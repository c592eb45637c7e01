/*
 * rcc.h -- C ABI of the MI355X-native calibration-target detection + pose hot path.
 *
 * This is the drop-in boundary for the per-frame path of Virtana/robot_camera_calibration
 * (SURVEY.md section 8(b)).  The reference has no function API for this path; it has
 *   (1) a ROS topic:  nh.subscribe("tag_detections", 1, aprilDetection)
 *         real_preprocessing/src/corner_detections.cpp:78, callback :41-65, which reads per
 *         detection  size[0] (:48), id[0] (:49), pixel_corners_x/y[0..3] (:53-54);
 *   (2) one solver call:  cv::solvePnP(obj_pts, img_pts, kcam_matrix, kdistCoeffs, rvec, tvec,
 *         false, CV_ITERATIVE)   real_preprocessing/src/camera_pose.cpp:163
 *       with the corner order bl,br,tr,tl (:152-155) and object points (+-size/2, z=0) (:158-161),
 *       followed by cv::Rodrigues (:164);
 *   (3) intrinsics as 9 row-major doubles + 5 plumb-bob doubles
 *         real_preprocessing/src/camera_pose.cpp:55-68.
 * Each entry point below names the reference interface it stands in for.
 *
 * Rules of the boundary: extern "C", plain pointers and sizes, POD structs with explicit sizes,
 * int status returns (0 = ok, <0 = error), nothing thrown across it, no torch types.
 * One handle per GPU per thread; handles are not internally locked; distinct handles are independent.
 */
#ifndef RCC_H_
#define RCC_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RCC_ABI_VERSION 2

/* ---- status codes -------------------------------------------------------------------------- */
enum {
  RCC_OK = 0,
  RCC_ERR_ARG = -1,          /* null pointer, bad size, bad enum */
  RCC_ERR_UNSUPPORTED = -2,  /* valid request this build does not implement */
  RCC_ERR_DEVICE = -3,       /* HIP runtime error; see rcc_last_device_error */
  RCC_ERR_CAPACITY = -4,     /* nframes / ntargets exceeds what the handle was created for */
  RCC_ERR_NOMEM = -5,
  RCC_ERR_STATE = -6         /* call out of order: submit with both result slots in flight, collect with nothing submitted */
};

/* ---- enums --------------------------------------------------------------------------------- */
/* sensor_msgs/Image encodings "mono8", "bgr8" (what cv_camera publishes, real_preprocessing/README.md:25) and "rgb8" (ABI 2):
 * the same luma Y = (1868 B + 9617 G + 4899 R + 8192) >> 14, with the byte order of the encoding */
enum { RCC_PIX_MONO8 = 0, RCC_PIX_BGR8 = 1, RCC_PIX_RGB8 = 2 };
/* distortion model of D[]: plumb-bob = (k1,k2,p1,p2,k3) as in camera_pose.cpp:39,64;
 * fisheye = (k1..k4), an extension the reference does not have. */
enum { RCC_DIST_NONE = 0, RCC_DIST_PLUMB_BOB = 1, RCC_DIST_FISHEYE = 2 };
enum { RCC_TARGET_CHECKERBOARD = 0, RCC_TARGET_FIDUCIAL = 1 };
enum { RCC_MEM_HOST = 0, RCC_MEM_DEVICE = 1 };
/* how the four corners of a square fiducial are refined (rcc_config.tag_refine):
 *   EDGES (default)  apriltag's refine_edges form: quads are found on coarsely localised corner
 *                    candidates (a5 cut to RCC_TAG_COARSE_ITERS iterations); per edge 16 samples search along the normal
 *                    for the gradient-weighted edge position, a total-least-squares line goes through them, and the
 *                    corners are the intersections of adjacent lines;
 *   CORNER_SUBPIX    every suppressed candidate goes through the cornerSubPix form of a5 first (the board's refinement,
 *                    run at an L-corner) and quads are found on the refined positions: rounds 1-2 of this build. */
enum { RCC_TAG_REFINE_EDGES = 0, RCC_TAG_REFINE_CORNER_SUBPIX = 1 };
/* EDGES: the a5 pass in front of the quad search is cut to a coarse localisation -- classification and linking work on
 * positions rounded to a pixel, the reported corners come from the edges */
#define RCC_TAG_COARSE_ITERS 2
#define RCC_TAG_COARSE_EPS   0.1

/* per-frame status word (bit flags) */
enum {
  RCC_FRAME_OK = 0,
  RCC_FRAME_CAND_OVERFLOW = 1, /* more Harris candidates than max_candidates: frame yields nothing */
  RCC_FRAME_NOT_FOUND = 2,     /* no complete target found */
  RCC_FRAME_KEPT_OVERFLOW = 4  /* more validated corners than max_kept: frame yields nothing */
};

/* per-target PnP status */
enum {
  RCC_PNP_OK = 0,
  RCC_PNP_TOO_FEW = 1,    /* fewer than 4 points */
  RCC_PNP_NONPLANAR = 2,  /* object points not coplanar: outside the reference's use (planar tags) */
  RCC_PNP_DEGENERATE = 3  /* non-finite homography; OpenCV falls back to R=I,t=0, so do we */
};

#define RCC_MAX_BOARD_CORNERS 256
#define RCC_MAX_KEPT_FIDUCIAL 2048   /* max_kept upper bound for RCC_TARGET_FIDUCIAL (256 for the board) */

/* ---- configuration (POD) ------------------------------------------------------------------- */
typedef struct rcc_config {
  uint32_t struct_size;   /* = sizeof(rcc_config); checked by rcc_create */
  uint32_t abi_version;   /* = RCC_ABI_VERSION */

  /* image geometry */
  int32_t width, height;
  int32_t stride_bytes;   /* bytes between rows of one input frame, >= width*channels */
  int32_t pixfmt;         /* RCC_PIX_* */
  int64_t frame_bytes;    /* bytes between consecutive frames of a batch, >= stride_bytes*height */

  /* intrinsics: K row-major 3x3 and D exactly as camera_pose.cpp:59-64 loads them */
  double K[9];
  int32_t dist_model;     /* RCC_DIST_* */
  int32_t undistort;      /* 1: per-pixel undistortion in the ingest pass, PnP then runs with D=0;
                             0: detector runs on the raw image, PnP uses D (the reference's way,
                                camera_pose.cpp:163) */
  double D[8];

  /* a3 adaptive threshold (apriltag tile min/max form) */
  int32_t thr_min_contrast; /* a tile whose 3x3-dilated max - min is below this is "flat" (127).  Default 16 (ABI 1: 32): apriltag's
                               own 5 (appendix B.3) turns the sensor noise of flat areas into salt and pepper */

  /* a4 corner extraction */
  int32_t harris_thresh;    /* accept R >= this (integer Harris response).  Default 10240 (ABI 1: 200000, which
                               loses the board under a Gaussian blur of sigma >= 1.5 px or 60 % shading: BASELINE.md section 4).  The
                               flat-tile skip of the threshold + corner pass is exact -- and only then taken -- while
                               (25 * g * g >> 4)^2 < harris_thresh with g = (4 * (thr_min_contrast - 1) + 7) >> 3: (16, 10240) and
                               (32, 200000) satisfy it; other pairs run the pass without the skip (same results, slower). */
  int32_t cand_margin;      /* candidates keep this many pixels from the image border (>= 8) */
  int32_t max_candidates;   /* per-frame capacity of the dense pass's output list */
  int32_t nms_radius;       /* list-level suppression radius (Chebyshev), pixels */
  int32_t xj_check;         /* 1: board scenes keep only X-junctions: a5 refines only candidates whose radius-11 grey ring shows >= 4
                               transitions, a4.3 validates on two radius-5 rings at the refined pixel; 0: every
                               suppressed candidate is refined and kept */
  int32_t max_kept;         /* board scenes: per-frame capacity of the VALIDATED list (entries that pass a4.3's ring tests; <= 256); the
                               list after suppression holds up to 2048 entries (ABI 1: this value bounded both, so ~100 objects in
                               view rejected the frame); tag scenes: capacity of the list after suppression (<= 2048).  More:
                               RCC_FRAME_KEPT_OVERFLOW, the frame yields nothing */

  /* a5 sub-pixel refinement (cornerSubPix form) */
  int32_t subpix_win;       /* half window w: (2w+1)^2 samples; 1..7 */
  int32_t subpix_max_iter;  /* 30 */
  double  subpix_eps;       /* stop when the step is shorter than this, pixels */

  /* a6 target */
  int32_t target_kind;      /* RCC_TARGET_* */
  int32_t board_cols, board_rows; /* inner corners, e.g. 8 x 6 (real_preprocessing/README.md:57) */
  double  board_square;     /* metres, e.g. 0.108 */
  int32_t board_id;         /* id reported for the board */
  int32_t max_targets;      /* result slots per frame */

  /* a7 */
  int32_t reference_mode;   /* 1: truncate sub-pixel corners to int before PnP, as
                               corner_detections.cpp:53-54 does */
  int32_t pnp_use_mfma;     /* 1: the 4-point tag poses (RCC_TARGET_FIDUCIAL) accumulate J^T J / J^T e with
                               v_mfma_f64_16x16x4_f64, two targets per instruction (csrc/k_pnp_mfma.hip); 0 (default):
                               vector FMAs, one lane per target -- the faster form, DESIGN.md section 5.  Same poses to 1e-9. */

  /* resources */
  int32_t device;           /* HIP device ordinal */
  int32_t batch_capacity;   /* max frames per rcc_detect_batch call */
  /* a6, fiducial targets (RCC_TARGET_FIDUCIAL): square tags of 8x8 cells -- a one-cell black border
   * around a 6x6 payload (white = 1), code word = payload row-major, MSB first.  The family table is
   * DATA supplied by the caller (the tag36h11 table is not redistributable from memory);
   * a build-generated family ships in robot_camera_calibration_amd/data/. */
  int32_t family_n;              /* number of codes */
  int32_t tag_max_hamming;       /* accept a decode with at most this many payload bit errors (2) */
  const uint64_t* family_codes;  /* host pointer to family_n code words; copied at rcc_create */
  double  tag_size;              /* metres, side of the black square: the four reported corners are its
                                    corners and the object points are (+-tag_size/2, +-tag_size/2, 0)
                                    (camera_pose.cpp:158-161) */
  int32_t tag_refine;            /* RCC_TAG_REFINE_*: corner refinement of fiducial quads (was reserved[0]: 0 = EDGES) */
  int32_t reserved[1];
} rcc_config;

/* ---- result records (POD) ------------------------------------------------------------------ */
/* One record per detected target: what apriltag_ros::AprilTagDetection carries to
 * corner_detections.cpp:46-56 (id, size, four pixel corners) plus the pose camera_pose.cpp:163-164
 * computes from them.  Corner order is bl, br, tr, tl (camera_pose.cpp:123-126,152-155). */
typedef struct rcc_detection {
  int32_t frame;          /* index within the batch */
  int32_t id;             /* targetID */
  int32_t hamming;        /* fiducial decode distance; 0 for the board */
  int32_t ncorners;       /* 4 for a fiducial, cols*rows for the board */
  double  size;           /* metres; side of the square spanned by corners[] in the object frame
                             (fiducial) or board_square (board) */
  double  corners[4][2];  /* bl, br, tr, tl pixel coordinates (sub-pixel; the reference casts to int) */
  double  rvec[3];        /* cam_T_target rotation vector, as solvePnP returns it */
  double  tvec[3];
  double  rms;            /* RMS reprojection error, pixels */
  int32_t pnp_status;     /* RCC_PNP_* */
  int32_t pnp_iters;
} rcc_detection;

/* Per-frame board output: all inner corners in row-major order of the board (row 0 first). */
typedef struct rcc_frame_corners {
  int32_t status;         /* RCC_FRAME_* flags */
  int32_t ncand;          /* dense-pass candidates */
  int32_t nkept;          /* after suppression + X-junction validation */
  int32_t ncorners;       /* 0 or cols*rows */
  int32_t px[RCC_MAX_BOARD_CORNERS][2]; /* integer candidate pixel (x,y): the bit-exact "corner index" */
  double  xy[RCC_MAX_BOARD_CORNERS][2]; /* refined sub-pixel position */
} rcc_frame_corners;

typedef struct rcc_handle rcc_handle;

/* ---- lifecycle ----------------------------------------------------------------------------- */
void rcc_default_config(rcc_config* cfg);  /* fills every field with the documented defaults */
int  rcc_create(const rcc_config* cfg, rcc_handle** out);
void rcc_destroy(rcc_handle* h);
const char* rcc_status_string(int status);
const char* rcc_last_device_error(const rcc_handle* h);
int  rcc_abi_version(void);

/* ---- the hot path -------------------------------------------------------------------------- */
/* Stands in for the external detector node that publishes "tag_detections"
 * (real_preprocessing/README.md:65; consumer corner_detections.cpp:41-56,78) plus the per-target
 * pose of camera_pose.cpp:163-164.
 *   frames     nframes images, frame_bytes apart, in host or device memory (frames_mem)
 *   det        out, capacity nframes*max_targets records, host memory; *ndet receives the count,
 *              records ordered by (frame, slot)
 *   corners    out, optional (may be NULL): nframes records, host memory
 *   stream     hipStream_t or NULL for the handle's own stream.  The handle's stream is a non-blocking one: it is NOT
 *              ordered after work the caller has queued elsewhere (not even on the null stream), so device inputs
 *              produced on another stream must be complete before the call, or the call must be given that stream
 * The call is synchronous with respect to its outputs. */
int rcc_detect_batch(rcc_handle* h, const void* frames, int32_t nframes, int32_t frames_mem,
                     rcc_detection* det, int32_t* ndet, rcc_frame_corners* corners, void* stream);

/* 1:1 stand-in for cv::solvePnP(obj, img, K, D, rvec, tvec, false, CV_ITERATIVE) at
 * camera_pose.cpp:163, batched over targets.  K/D NULL means the handle's.
 *   obj    sum(npts) x 3 doubles, img  sum(npts) x 2 doubles, targets concatenated
 *   npts   ntargets counts
 *   rvec, tvec  ntargets x 3; rms, status, iters ntargets (rms/status/iters may be NULL)
 * All pointers host memory. */
int rcc_solve_pnp_batch(rcc_handle* h, const double* obj, const double* img, const int32_t* npts,
                        int32_t ntargets, const double* K, const double* D, int32_t dist_model,
                        double* rvec, double* tvec, double* rms, int32_t* status, int32_t* iters);

/* cv::Rodrigues both ways (camera_pose.cpp:93,116,164; opt_visualization.cpp:36), batched, host
 * pointers; computed on the device. */
int rcc_rodrigues_v2m_batch(rcc_handle* h, const double* rvec, int32_t n, double* R9);
int rcc_rodrigues_m2v_batch(rcc_handle* h, const double* R9, int32_t n, double* rvec);

/* ---- stage-level entry points (device pointers; used by the parity tests and the bench) ------ */
/* Asynchronous form of rcc_detect_batch for a stream of batches: submit launches everything for a batch (incl.
 * the copy of the records into one of two pinned result slots) and returns; collect waits for the OLDEST
 * outstanding submission and unpacks its records (same order and contents as rcc_detect_batch).  At most two
 * submissions may be outstanding, so the host can unpack batch k while the device runs batch k+1:
 *     submit(b0); for (k = 1; k < n; ++k) { submit(b_k); collect(&out[k-1]); }  collect(&out[n-1]);
 * `frames` (and `corners`, if given) must stay valid until that submission has been collected; `corners` is complete when
 * its collect returns (the tables come over on a copy stream under the next batch's first passes, into a pinned slot of the
 * handle -- the caller's array may be pageable -- and collect copies them out).  RCC_ERR_STATE when called out of order; rcc_detect_batch itself refuses to run
 * while submissions are outstanding. */
int rcc_detect_batch_submit(rcc_handle* h, const void* frames, int32_t nframes, int32_t frames_mem,
                            rcc_frame_corners* corners, void* stream);
int rcc_detect_batch_collect(rcc_handle* h, rcc_detection* det, int32_t* ndet);

/* ---- records for the exchange between processes (multi-GPU) -------------------------------------
 * Between processes the reference hands detections on over the "tag_detections" topic (corner_detections.cpp:78); with
 * one process per GPU the per-batch stand-in is ONE all-gather of fixed-size records.  rcc_set_record_tables makes
 * every following rcc_detect_batch / rcc_detect_batch_submit also write, on the device and on the batch's stream, a
 * table of RCC_REC_DOUBLES doubles per slot into the caller's DEVICE buffer (rcc_record_slots(nframes) slots; slot =
 * frame * targets_per_frame + q, all zeros where there is no target):
 *   [0] valid (1.0)  [1] frame_offset + frame  [2] id  [3] ncorners  [4..6] rvec  [7..9] tvec  [10] rms
 *   [11..18] the four corners bl, br, tr, tl as x,y -- all four, as the consumer reads them (corner_detections.cpp:51-56)
 * d_table0 serves rcc_detect_batch and the submissions in result slot 0, d_table1 those in slot 1 (NULL: d_table0 for
 * both -- then a table must be consumed before the next submission).  A submission's table is complete when its
 * rcc_detect_batch_collect returns.  NULL, NULL switches the tables off.  include/rcc_dist.h gathers such tables over RCCL.
 * capacity_slots = slots each table holds (>= 1 when a table is given): a batch whose rcc_record_slots(nframes) exceeds it
 * is refused with RCC_ERR_CAPACITY before anything is launched, and a SHORTER batch (a ragged last one) leaves the slots
 * [rcc_record_slots(nframes), capacity_slots) all zeros -- a gather of the whole table never sees a previous batch's records. */
#define RCC_REC_DOUBLES 19
int rcc_set_record_tables(rcc_handle* h, double* d_table0, double* d_table1, int32_t capacity_slots, int32_t frame_offset);
int rcc_record_slots(const rcc_handle* h, int32_t nframes);

/* a1+a2: ingest = BGR->grey (+ undistort when cfg.undistort).  src/dst device pointers;
 * dst is nframes tightly packed width*height u8 images. */
int rcc_stage_ingest(rcc_handle* h, const void* d_frames, int32_t nframes, void* d_grey, void* stream);
/* a3+a4 dense pass: grey -> threshold map {0,127,255} + candidate list.
 *   d_bin       nframes*width*height u8
 *   d_cand      nframes*max_candidates entries {int16 x, int16 y, int32 score}, unordered
 *   d_cand_count nframes int32 (true count, may exceed max_candidates) */
int rcc_stage_threshold_corner(rcc_handle* h, const void* d_grey, int32_t nframes, void* d_bin,
                               void* d_cand, void* d_cand_count, void* stream);
/* rcc_detect_batch and the binary image: the stages after the threshold+corner pass read it at 16 ring points per
 * corner only, so by default the pass leaves it as a compact map (one byte per 4x4 tile: level or "flat") -- 1/16 of
 * the output bytes, same decisions -- and rcc_debug_fetch_images expands it on demand.  on = 1 makes
 * rcc_detect_batch materialise the full binary image (as rcc_stage_threshold_corner always does).  Returns the
 * previous setting. */
int rcc_set_keep_binary(rcc_handle* h, int on);
/* switches rcc_config.pnp_use_mfma of a live handle (A/B timing, tests).  Returns the previous setting. */
int rcc_set_pnp_mfma(rcc_handle* h, int on);
/* a4 list stage + a5 + a6 + a7 for the board: consumes the dense pass outputs, fills per-frame
 * device records (layout = rcc_frame_corners / rcc_detection), then copies to host. */
int rcc_stage_targets(rcc_handle* h, const void* d_grey, const void* d_bin, const void* d_cand,
                      const void* d_cand_count, int32_t nframes, rcc_detection* det, int32_t* ndet,
                      rcc_frame_corners* corners, void* stream);

/* Test taps, HIP-event timers and A/B switches between bit-identical kernel variants are NOT part of this boundary:
 * they are declared in include/rcc_debug.h (test + measurement infrastructure). */

/* ---- synthetic camera (stands where rviz_simulator's missing camera.h was meant to be,
 *      rviz_simulator/include/rviz_simulator/target.h:40) --------------------- */
typedef struct rcc_synth_params {
  uint32_t struct_size;
  int32_t  board_cols, board_rows;  /* inner corners */
  double   board_square;
  int32_t  margin_squares;          /* white quiet zone around the board, in squares (1) */
  int32_t  supersample;             /* s x s samples per pixel (4) */
  double   noise_sigma;             /* additive Gaussian noise, LSB (2) */
  uint64_t seed;                    /* frame f uses seed + f (0xC0FFEE) */
  int32_t  black, white, background;/* 20, 235, 128 */
  int32_t  fid_grid_x, fid_grid_y;  /* > 0: render a planar grid of fiducials (ids 0..) instead of the board */
  int32_t  fid_gap_permille;        /* white gap between tags, in 1/1000 of the tag size (500) */
  int32_t  reserved[2];
  /* ---- optics (ABI 2).  All zero = the ideal camera of ABI 1, bit for bit.  The real input is a webcam through cv_camera
   * (real_preprocessing/README.md:25,64-65): its images are not razor-edged.  Everything here is INTEGER arithmetic on the
   * supersampled image (the sum of the s x s integer samples of a pixel), applied before the sensor noise:
   *   blur:      separable, symmetric; blur_taps[k] = weight at distance k (k = 0..RCC_SYNTH_BLUR_TAPS-1), and
   *              blur_taps[0] + 2 * sum(blur_taps[1..]) must be 256 (or all zero: no blur); rows and columns are clamped at the
   *              image border.  {128, 64} is the 3-tap 1-2-1 filter; the Python mirror (abi.py, gaussian_taps) fills a Gaussian.
   *   shading:   gain(u, v) in 1/4096: lin = 4096 + trunc(4096 * (gx * X + gy * Y) / 1000), X = (2u - (w-1)) / (w-1) in [-1, 1]
   *              (shade_x_permille = gx: relative brightness change from the centre to the right edge), Y alike;
   *              vig = 4096 - trunc(4096 * vignette_permille * r2 / (1000 * R2)), r2 = (2u-(w-1))^2 + (2v-(h-1))^2, R2 its value in
   *              a corner (vignette_permille = relative darkening of the corners); gain = lin * vig >> 12.
   *   pixel = rint(blurred * gain / (256 * 256 * 4096 * s * s) + noise), clamped to [0, 255]. */
  int32_t  blur_taps[8];
  int32_t  shade_x_permille, shade_y_permille;
  int32_t  vignette_permille;
  int32_t  reserved2[1];
} rcc_synth_params;
#define RCC_SYNTH_BLUR_TAPS 8

/* Render nframes frames of the handle's geometry/intrinsics into d_frames (device), one pose per
 * frame: poses = nframes x 6 doubles (rvec, tvec of cam_T_target), host memory. */
int rcc_synth_render_batch(rcc_handle* h, const rcc_synth_params* sp, const double* poses,
                           int32_t nframes, int32_t first_frame_index, void* d_frames, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RCC_H_ */
