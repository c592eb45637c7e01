/*
 * orc.h -- CPU ORACLE for the calibration-target detection + pose hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only as the checker.
 *
 * PARITY UNPINNED: the reference (Virtana/robot_camera_calibration) holds no tests, fixtures or
 * golden vectors for this path, its pixel stages live in un-vendored third-party packages
 * (apriltag_ros fork @ milestone_1b_pipeline, apriltag unpinned, OpenCV 3.4.4 -- SURVEY.md 8(c)),
 * and nothing of it can be built or run in this image.  This restatement follows
 *   - the reference's own call sites and conventions, cited file:line at each function, and
 *   - the published algorithms of those dependencies as restated in SURVEY.md appendix A-C.
 * It therefore DEFINES the behaviour the HIP path is checked against; it is pinned only by
 * analytic known-answer tests (tests/test_oracle_*.py), not by reference outputs.
 *
 * Build: gcc -O2 -ffp-contract=off (see oracle/Makefile).  -ffp-contract=off matters: the float
 * stages are specified as sequences of individually rounded IEEE-754 operations so that a device
 * implementation issuing the same operations reproduces them bit for bit.
 */
#ifndef ORC_H_
#define ORC_H_

#include <stdint.h>
#include "../include/rcc.h"   /* POD structs of the boundary contract only */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_cand { int16_t x, y; int32_t score; } orc_cand;

/* ---- a1 / a2 ingest ---- */
void orc_bgr_to_grey(const uint8_t* bgr, int w, int h, int stride, uint8_t* grey);
void orc_rgb_to_grey(const uint8_t* rgb, int w, int h, int stride, uint8_t* grey);
double orc_atan_pos(double r);
void orc_undistort_map_q5(const double K[9], int dist_model, const double D[8], int w, int h,
                          int32_t* mapx, int32_t* mapy);
void orc_remap_q5(const uint8_t* src, int w, int h, int stride, const int32_t* mapx,
                  const int32_t* mapy, uint8_t* dst);
/* full ingest of one frame as cfg describes it (grey, then optional undistortion) */
int orc_ingest(const rcc_config* cfg, const uint8_t* frame, uint8_t* grey_out);

/* ---- a3 threshold ---- */
void orc_threshold_tiles(const uint8_t* grey, int w, int h, int min_contrast, uint8_t* bin);

/* ---- a4 corners ---- */
void orc_harris_response(const uint8_t* grey, int w, int h, int32_t* R);
/* 3x3 local maxima above thresh inside margin, sorted by (y,x). returns the true count,
 * writes at most cap entries */
int orc_harris_candidates(const int32_t* R, int w, int h, int thresh, int margin, orc_cand* out, int cap);
/* list-level radius suppression + X-junction ring validation; in sorted by (y,x); returns kept count */
int orc_filter_candidates(const orc_cand* in, int n, const uint8_t* bin, int w, int h,
                          int nms_radius, int xj_check, orc_cand* out, int cap);
int orc_xjunction_ring(const uint8_t* bin, int w, int h, int x, int y);
int orc_xjunction_ring_grey(const uint8_t* grey, int w, int h, int x, int y, int min_contrast);
int orc_junction_pretest(const uint8_t* grey, int w, int h, int x, int y, int min_contrast);

/* ---- a5 sub-pixel ---- */
void orc_corner_subpix(const uint8_t* grey, int w, int h, const orc_cand* pts, int n,
                       int win, int max_iter, double eps, double* xy_out);

/* a4.3: ring test + de-duplication at the rounded refined position; keeps input order */
int orc_validate_refined(const orc_cand* pre, int n, const double* xy, const uint8_t* bin, const uint8_t* grey, int w,
                         int h, int xj_check, int min_contrast, int dedupe_radius, orc_cand* out, double* out_xy, int cap);

/* ---- a6 board indexing ---- */
/* pts: kept candidates sorted by (y,x). order_out[cols*rows] receives indices into pts in
 * row-major board order. returns 1 if the board was found */
int orc_grid_index(const orc_cand* pts, int n, int cols, int rows, int32_t* order_out);

/* ---- a4/a6 square-fiducial form (orc_fiducial.c) ---- */
int orc_fid_corner_class(const uint8_t* g, int w, int h, int x, int y, int min_contrast, int d1[2], int d2[2], int* thr);
int orc_fid_homography(const double q[8], double H[9]);
int orc_fid_decode(const uint8_t* g, int w, int h, const double q[8], const uint64_t* codes, int ncodes,
                   int max_hamming, int* id_out, int* ham_out, int* rot_out);
int orc_fid_detect(const uint8_t* grey, int w, int h, int min_contrast, const orc_cand* pre, const double* xy,
                   int n, const uint64_t* codes, int ncodes, int max_hamming, int refine_mode, rcc_detection* out, int cap);
void orc_fid_refine_edges(const uint8_t* g, int w, int h, const int qi[8], double qr[8]);

/* ---- a7 / a8 pose ---- */
void orc_rodrigues_v2m(const double r[3], double R[9], double J[27]);  /* J: 3x9, may be NULL */
void orc_rodrigues_m2v(const double R[9], double r[3]);
void orc_undistort_points(const double* img, int n, const double K[9], int dist_model,
                          const double D[8], double* out);
void orc_project_points(const double* obj, int n, const double r[3], const double t[3],
                        const double K[9], int dist_model, const double D[8], double* uv,
                        double* dpdr /* 2n x 3 or NULL */, double* dpdt /* 2n x 3 or NULL */);
int orc_find_homography(const double* src, const double* dst, int n, double H[9]);
int orc_solve_pnp(const double* obj, const double* img, int n, const double K[9], int dist_model,
                  const double D[8], double rvec[3], double tvec[3], double* rms, int* iters);
void orc_jacobi_eigen_sym(int n, double* A /* n*n, destroyed */, double* w, double* V /* rows = vectors */);

/* ---- whole path for one frame ---- */
/* scratch-free convenience: allocates what it needs. Returns number of detections (0 or 1 for
 * the board). */
int orc_detect_frame(const rcc_config* cfg, const uint8_t* frame, int frame_index,
                     rcc_detection* det, rcc_frame_corners* fc);
/* same, with the stage outputs exposed for the parity tests (any pointer may be NULL) */
int orc_detect_frame_ex(const rcc_config* cfg, const uint8_t* frame, int frame_index,
                        rcc_detection* det, rcc_frame_corners* fc, uint8_t* grey_out,
                        uint8_t* bin_out, orc_cand* cand_out, int32_t* ncand_out,
                        orc_cand* pre_out, int32_t* npre_out, double* pre_xy_out,
                        orc_cand* kept_out, int32_t* nkept_out);

/* ---- synthetic camera ---- */
/* returns 0, -1 for malformed optics parameters (include/rcc.h: blur taps must sum to 256), -2 out of memory */
int orc_synth_render(const rcc_config* cfg, const rcc_synth_params* sp, const double pose[6],
                     int frame_index, uint8_t* frame_out);
int orc_synth_optics(const rcc_synth_params* sp, int taps[RCC_SYNTH_BLUR_TAPS]);
void orc_board_object_points(int cols, int rows, double square, double* obj /* cols*rows*3 */);

#ifdef __cplusplus
}
#endif
#endif
