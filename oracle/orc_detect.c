/*
 * orc_detect.c -- CPU oracle, the whole per-frame path a1..a10 chained.  TEST INFRASTRUCTURE.
 *
 * Output contract: what the reference consumes per detection is id[0], size[0]
 * and four pixel corners (real_preprocessing/src/corner_detections.cpp:48-54); the pose is what
 * camera_pose.cpp:163-164 would compute from them.  reference_mode reproduces the int() cast of
 * corner_detections.cpp:53-54 before the solve.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "orc.h"

typedef struct orc_ctx {
  rcc_config cfg;
  int32_t* mapx;
  int32_t* mapy;
  uint8_t* tmp;
  uint8_t* grey;
  uint8_t* bin;
  int32_t* R;
  orc_cand* cand;
  orc_cand* pre;
  orc_cand* kept;
  double* pxy;
} orc_ctx;

void orc_ctx_destroy(orc_ctx* c)
{
  if (!c) return;
  free(c->mapx); free(c->mapy); free(c->tmp); free(c->grey); free(c->bin); free(c->R);
  free(c->cand); free(c->pre); free(c->kept); free(c->pxy);
  free(c);
}

orc_ctx* orc_ctx_create(const rcc_config* cfg)
{
  orc_ctx* c = (orc_ctx*)calloc(1, sizeof(orc_ctx));
  if (!c) return NULL;
  c->cfg = *cfg;
  size_t n = (size_t)cfg->width * cfg->height;
  c->tmp = (uint8_t*)malloc(n);
  c->grey = (uint8_t*)malloc(n);
  c->bin = (uint8_t*)malloc(n);
  c->R = (int32_t*)malloc(n * sizeof(int32_t));
  c->cand = (orc_cand*)malloc(sizeof(orc_cand) * (size_t)(cfg->max_candidates > 0 ? cfg->max_candidates : 1));
  c->pre = (orc_cand*)malloc(sizeof(orc_cand) * RCC_MAX_KEPT_FIDUCIAL);
  c->kept = (orc_cand*)malloc(sizeof(orc_cand) * 256);
  c->pxy = (double*)malloc(sizeof(double) * 2 * RCC_MAX_KEPT_FIDUCIAL);
  if (cfg->undistort && cfg->dist_model != RCC_DIST_NONE) {
    c->mapx = (int32_t*)malloc(n * sizeof(int32_t));
    c->mapy = (int32_t*)malloc(n * sizeof(int32_t));
    if (c->mapx && c->mapy) orc_undistort_map_q5(cfg->K, cfg->dist_model, cfg->D, cfg->width, cfg->height, c->mapx, c->mapy);
  }
  if (!c->tmp || !c->grey || !c->bin || !c->R || !c->cand || !c->pre || !c->kept || !c->pxy) { orc_ctx_destroy(c); return NULL; }
  return c;
}

static void ingest(orc_ctx* c, const uint8_t* frame)
{
  const rcc_config* cfg = &c->cfg;
  const int w = cfg->width, h = cfg->height;
  uint8_t* g0 = c->mapx ? c->tmp : c->grey;
  if (cfg->pixfmt == RCC_PIX_BGR8) orc_bgr_to_grey(frame, w, h, cfg->stride_bytes, g0);
  else if (cfg->pixfmt == RCC_PIX_RGB8) orc_rgb_to_grey(frame, w, h, cfg->stride_bytes, g0);
  else for (int y = 0; y < h; ++y) memcpy(g0 + (size_t)y * w, frame + (size_t)y * cfg->stride_bytes, w);
  if (c->mapx) orc_remap_q5(c->tmp, w, h, w, c->mapx, c->mapy, c->grey);
}

int orc_ctx_detect(orc_ctx* c, const uint8_t* frame, int frame_index, rcc_detection* det,
                   rcc_frame_corners* fc_out, uint8_t* grey_out, uint8_t* bin_out,
                   orc_cand* cand_out, int32_t* ncand_out, orc_cand* pre_out, int32_t* npre_out,
                   double* pre_xy_out, orc_cand* kept_out, int32_t* nkept_out)
{
  const rcc_config* cfg = &c->cfg;
  const int w = cfg->width, h = cfg->height;
  rcc_frame_corners fcl;
  rcc_frame_corners* fc = fc_out ? fc_out : &fcl;
  memset(fc, 0, sizeof(*fc));
  /* capacities [B].  The list after suppression (a4.2) holds up to RCC_MAX_KEPT_FIDUCIAL = 2048 entries for every target kind (tag
   * scenes: cfg->max_kept of them at most); for the board cfg->max_kept (<= 256, what the lattice stage is built for) bounds the
   * VALIDATED list of a4.3 only -- since round 4: a5's gate drops what cannot be a junction before it costs anything, so a cluttered
   * scene may bring hundreds of suppressed candidates (rounds 1-3 rejected the frame beyond 256). */
  const int fidt = (cfg->target_kind == RCC_TARGET_FIDUCIAL);
  const int max_pre = fidt ? (cfg->max_kept < RCC_MAX_KEPT_FIDUCIAL ? cfg->max_kept : RCC_MAX_KEPT_FIDUCIAL) : RCC_MAX_KEPT_FIDUCIAL;
  const int max_kept = fidt ? max_pre : (cfg->max_kept < 256 ? cfg->max_kept : 256);

  ingest(c, frame);
  orc_threshold_tiles(c->grey, w, h, cfg->thr_min_contrast, c->bin);
  orc_harris_response(c->grey, w, h, c->R);
  int ncand = orc_harris_candidates(c->R, w, h, cfg->harris_thresh, cfg->cand_margin, c->cand, cfg->max_candidates);
  if (grey_out) memcpy(grey_out, c->grey, (size_t)w * h);
  if (bin_out) memcpy(bin_out, c->bin, (size_t)w * h);
  if (ncand_out) *ncand_out = ncand;
  if (cand_out) memcpy(cand_out, c->cand, sizeof(orc_cand) * (size_t)(ncand < cfg->max_candidates ? ncand : cfg->max_candidates));
  if (npre_out) *npre_out = 0;
  if (nkept_out) *nkept_out = 0;
  fc->ncand = ncand;
  if (ncand > cfg->max_candidates) { fc->status = RCC_FRAME_CAND_OVERFLOW; return 0; }

  /* a4.2 list-level suppression (no ring test yet), a5 refine every survivor, a4.3 validate */
  int npre = orc_filter_candidates(c->cand, ncand, c->bin, w, h, cfg->nms_radius, 0, c->pre, max_pre);
  if (npre_out) *npre_out = npre > max_pre ? 0 : npre;
  if (npre > max_pre) { fc->status = RCC_FRAME_KEPT_OVERFLOW; return 0; }
  if (pre_out) memcpy(pre_out, c->pre, sizeof(orc_cand) * (size_t)npre);
  double* pxy = c->pxy;
  double xy[2 * 256];
  if (cfg->target_kind == RCC_TARGET_FIDUCIAL) {
    /* Square fiducials: only what can become a quad corner is refined.  A candidate whose own pixel does not pass the
     * convex-black-corner test (orc_fid_corner_class, the test the quad stage applies to the rounded refined position) keeps
     * that pixel as its position -- where the quad stage's test fails for it again, so it can neither be a corner nor
     * attract a link.  About two thirds of a tag scene's candidates (the payload's inner corners) go this way.
     * refine_edges form: the a5 pass of the others only has to bring them within a pixel of their corner (classification
     * and linking work on rounded positions; the reported corners come from the edges): at most RCC_TAG_COARSE_ITERS
     * iterations, stop below RCC_TAG_COARSE_EPS px.  cornerSubPix form: the configured iteration limits. */
    const int tag_edges = (cfg->tag_refine == RCC_TAG_REFINE_EDGES);
    const int it = (tag_edges && cfg->subpix_max_iter > RCC_TAG_COARSE_ITERS) ? RCC_TAG_COARSE_ITERS : cfg->subpix_max_iter;
    const double eps = (tag_edges && cfg->subpix_eps < RCC_TAG_COARSE_EPS) ? RCC_TAG_COARSE_EPS : cfg->subpix_eps;
    for (int i = 0; i < npre; ++i) {
      int a[2], b[2], t;
      if (orc_fid_corner_class(c->grey, w, h, c->pre[i].x, c->pre[i].y, cfg->thr_min_contrast, a, b, &t))
        orc_corner_subpix(c->grey, w, h, c->pre + i, 1, cfg->subpix_win, it, eps, pxy + 2 * i);
      else { pxy[2 * i] = (double)c->pre[i].x; pxy[2 * i + 1] = (double)c->pre[i].y; }
    }
  } else if (cfg->xj_check && cfg->target_kind == RCC_TARGET_CHECKERBOARD) {
    /* board scenes: only what can be a junction is refined (orc_junction_pretest); the others reach a4.3 as (-1, -1) */
    for (int i = 0; i < npre; ++i) {
      if (orc_junction_pretest(c->grey, w, h, c->pre[i].x, c->pre[i].y, cfg->thr_min_contrast))
        orc_corner_subpix(c->grey, w, h, c->pre + i, 1, cfg->subpix_win, cfg->subpix_max_iter, cfg->subpix_eps, pxy + 2 * i);
      else { pxy[2 * i] = -1.0; pxy[2 * i + 1] = -1.0; }
    }
  } else {
    orc_corner_subpix(c->grey, w, h, c->pre, npre, cfg->subpix_win, cfg->subpix_max_iter, cfg->subpix_eps, pxy);
  }
  if (pre_xy_out) memcpy(pre_xy_out, pxy, sizeof(double) * 2 * (size_t)npre);
  if (cfg->target_kind == RCC_TARGET_FIDUCIAL) {
    /* a4/a6 square-fiducial form + a7 per tag (4 corners bl,br,tr,tl; camera_pose.cpp:152-163) */
    const int cap = cfg->max_targets;
    rcc_detection* tmp = (rcc_detection*)malloc(sizeof(rcc_detection) * (size_t)(cap > 0 ? cap : 1));
    int m = orc_fid_detect(c->grey, w, h, cfg->thr_min_contrast, c->pre, pxy, npre, cfg->family_codes, cfg->family_n,
                           cfg->tag_max_hamming, cfg->tag_refine, tmp, cap);
    if (m > cap) m = cap;
    const double s2 = 0.5 * cfg->tag_size;
    const double obj[12] = { -s2, -s2, 0, s2, -s2, 0, s2, s2, 0, -s2, s2, 0 };
    const int undist = cfg->undistort && cfg->dist_model != RCC_DIST_NONE;
    for (int k = 0; k < m; ++k) {
      rcc_detection* d = tmp + k;
      d->frame = frame_index;
      d->size = cfg->tag_size;
      double img[8];
      for (int q = 0; q < 4; ++q) {
        double x = d->corners[q][0], y = d->corners[q][1];
        if (cfg->reference_mode) { x = (double)(int)x; y = (double)(int)y; }
        img[2 * q] = x; img[2 * q + 1] = y;
      }
      int iters = 0;
      d->pnp_status = orc_solve_pnp(obj, img, 4, cfg->K, undist ? RCC_DIST_NONE : cfg->dist_model, cfg->D,
                                    d->rvec, d->tvec, &d->rms, &iters);
      d->pnp_iters = iters;
      if (det) det[k] = *d;
    }
    free(tmp);
    fc->nkept = npre;
    if (m == 0) fc->status = RCC_FRAME_NOT_FOUND;
    return m;
  }
  int nkept = orc_validate_refined(c->pre, npre, pxy, c->bin, c->grey, w, h, cfg->xj_check, cfg->thr_min_contrast, 2, c->kept, xy, max_kept);
  if (nkept > max_kept) { fc->status = RCC_FRAME_KEPT_OVERFLOW; return 0; }       /* more validated points than the lattice stage takes */
  fc->nkept = nkept;
  if (nkept_out) *nkept_out = nkept;
  if (kept_out) memcpy(kept_out, c->kept, sizeof(orc_cand) * (size_t)nkept);

  if (cfg->target_kind != RCC_TARGET_CHECKERBOARD) { fc->status = RCC_FRAME_NOT_FOUND; return 0; }
  const int nc = cfg->board_cols * cfg->board_rows;
  int32_t order[RCC_MAX_BOARD_CORNERS];
  if (nc > RCC_MAX_BOARD_CORNERS || !orc_grid_index(c->kept, nkept, cfg->board_cols, cfg->board_rows, order)) {
    fc->status = RCC_FRAME_NOT_FOUND;
    return 0;
  }
  fc->ncorners = nc;
  for (int k = 0; k < nc; ++k) {
    fc->px[k][0] = c->kept[order[k]].x;
    fc->px[k][1] = c->kept[order[k]].y;
    fc->xy[k][0] = xy[2 * order[k]];
    fc->xy[k][1] = xy[2 * order[k] + 1];
  }
  if (!det) return 1;

  /* a7: object points as camera_pose.cpp:158-161 lays them out (x right, y up, z = 0, centred) */
  double obj[3 * RCC_MAX_BOARD_CORNERS], img[2 * RCC_MAX_BOARD_CORNERS];
  orc_board_object_points(cfg->board_cols, cfg->board_rows, cfg->board_square, obj);
  for (int k = 0; k < nc; ++k) {
    double x = fc->xy[k][0], y = fc->xy[k][1];
    if (cfg->reference_mode) { x = (double)(int)x; y = (double)(int)y; }  /* corner_detections.cpp:53-54 */
    img[2 * k] = x; img[2 * k + 1] = y;
  }
  memset(det, 0, sizeof(*det));
  det->frame = frame_index;
  det->id = cfg->board_id;
  det->hamming = 0;
  det->ncorners = nc;
  det->size = cfg->board_square;
  /* bl, br, tr, tl of the inner-corner lattice (camera_pose.cpp:123-126 order) */
  const int C = cfg->board_cols, Rr = cfg->board_rows;
  const int idx[4] = { (Rr - 1) * C, (Rr - 1) * C + C - 1, C - 1, 0 };
  for (int k = 0; k < 4; ++k) { det->corners[k][0] = fc->xy[idx[k]][0]; det->corners[k][1] = fc->xy[idx[k]][1]; }
  const int undist = cfg->undistort && cfg->dist_model != RCC_DIST_NONE;
  int iters = 0;
  det->pnp_status = orc_solve_pnp(obj, img, nc, cfg->K, undist ? RCC_DIST_NONE : cfg->dist_model,
                                  cfg->D, det->rvec, det->tvec, &det->rms, &iters);
  det->pnp_iters = iters;
  return 1;
}

int orc_detect_frame_ex(const rcc_config* cfg, const uint8_t* frame, int frame_index,
                        rcc_detection* det, rcc_frame_corners* fc, uint8_t* grey_out,
                        uint8_t* bin_out, orc_cand* cand_out, int32_t* ncand_out,
                        orc_cand* pre_out, int32_t* npre_out, double* pre_xy_out,
                        orc_cand* kept_out, int32_t* nkept_out)
{
  orc_ctx* c = orc_ctx_create(cfg);
  if (!c) return RCC_ERR_NOMEM;
  int r = orc_ctx_detect(c, frame, frame_index, det, fc, grey_out, bin_out, cand_out, ncand_out, pre_out, npre_out, pre_xy_out, kept_out, nkept_out);
  orc_ctx_destroy(c);
  return r;
}

int orc_detect_frame(const rcc_config* cfg, const uint8_t* frame, int frame_index,
                     rcc_detection* det, rcc_frame_corners* fc)
{
  return orc_detect_frame_ex(cfg, frame, frame_index, det, fc, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL);
}

/* bench.py's cpu_baseline leg: run the whole path over nframes frames with one context.
 * Returns the number of frames in which the board was found. */
int orc_ctx_detect_many(orc_ctx* c, const uint8_t* frames, int64_t frame_bytes, int nframes,
                        rcc_detection* det_out)
{
  int found = 0;
  for (int f = 0; f < nframes; ++f) {
    rcc_detection d[64];
    int r = orc_ctx_detect(c, frames + (size_t)f * frame_bytes, f, c->cfg.max_targets <= 64 ? d : NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL);
    if (r > 0) {
      if (det_out && c->cfg.max_targets <= 64) det_out[found] = d[0];
      ++found;
    }
  }
  return found;
}

/* default configuration shared by the oracle-side tests; the product has its own
 * rcc_default_config and a test checks the two agree field by field */
void orc_default_config(rcc_config* c)
{
  memset(c, 0, sizeof(*c));
  c->struct_size = sizeof(rcc_config);
  c->abi_version = RCC_ABI_VERSION;
  c->width = 640; c->height = 480; c->stride_bytes = 640 * 3; c->pixfmt = RCC_PIX_BGR8;
  c->frame_bytes = (int64_t)640 * 3 * 480;
  c->K[0] = c->K[4] = 0.9 * 640; c->K[2] = (640 - 1) * 0.5; c->K[5] = (480 - 1) * 0.5; c->K[8] = 1.0;
  c->dist_model = RCC_DIST_PLUMB_BOB;
  c->undistort = 1;
  c->D[0] = -0.28; c->D[1] = 0.07; c->D[2] = 2e-4; c->D[3] = -1e-4; c->D[4] = 0.0;
  c->thr_min_contrast = 16;      /* (16, 10240) since round 4: include/rcc.h */
  c->harris_thresh = 10240;
  c->cand_margin = 8;
  c->max_candidates = 2048;
  c->nms_radius = 5;
  c->xj_check = 1;
  c->max_kept = 256;
  c->subpix_win = 5;
  c->subpix_max_iter = 30;
  c->subpix_eps = 1e-3;
  c->target_kind = RCC_TARGET_CHECKERBOARD;
  c->board_cols = 8; c->board_rows = 6; c->board_square = 0.108; c->board_id = 0;
  c->max_targets = 1;
  c->reference_mode = 0;
  c->pnp_use_mfma = 0;
  c->device = 0;
  c->batch_capacity = 16;
}
