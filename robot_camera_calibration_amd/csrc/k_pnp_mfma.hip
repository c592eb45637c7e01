// k_pnp_mfma.hip -- the 4-point tag pose (stage a7, cv::solvePnP at real_preprocessing/src/camera_pose.cpp:163 with the
// object points of :158-161) with the normal equations of the Levenberg-Marquardt step accumulated on the matrix cores:
// rcc_config.pnp_use_mfma = 1 (BASELINE.json north_star: "MFMA only for the small batched JtJ/Jtr normal-equation
// blocks"; configs[4]).  Same solver as k_pnp_tags (k_fid.hip): same initial pose (pnp_core.h pose_init), same
// CvLevMarq schedule, same Cholesky solve -- only J^T J, J^T e and |e|^2 come from v_mfma_f64_16x16x4_f64.
//
// Formulation.  Per target G = [J | e] is 8 x 7 (two rows per corner, six pose parameters + the residual), and
// G^T G holds everything the step needs: J^T J (6 x 6), J^T e (column 6), |e|^2 (entry 6,6).  One MFMA computes a
// 16 x 16 product of depth 4, so TWO targets share an instruction: the A operand is [G_a | G_b]^T (16 x 4: one half of the
// 8 rows), the B operand its transpose -- which, in the instruction's register layout (A[l & 15][l >> 4], B[l >> 4][l & 15]),
// is the SAME register: each lane loads one double and passes it twice.  Two instructions (rows 0-3, 4-7) finish a pair;
// the two 8 x 8 diagonal blocks of the result are G_a^T G_a and G_b^T G_b, the off-diagonal blocks (a's Jacobian against
// b's) are discarded: 98 of 256 results are used.
//
// A wavefront holds 64 targets, one per lane (as k_pnp_tags).  MFMA reads whole wavefronts, so the LM loops of the 64
// lanes run in lock-step ROUNDS instead of diverging: a round = [lanes that need new normal equations write G to LDS]
// -> 32 pairs x 2 MFMA -> [those lanes read A, g, |e|^2 back] -> every unfinished lane solves with its current damping
// and evaluates the trial pose.  A rejected trial only raises the damping: that lane sits out the next round's G phase.
//
// Measured against the vector form: profiles/r02_*_pnp_mfma.txt, DESIGN.md section 5.
#include "rcc_internal.h"
// The solver routines of pnp_core.h are inlined (no RCC_PNP_NOINLINE): round 1 kept them out of line after a suspected
// hipcc -O3 miscompile that round 2 could not reproduce -- the fully inlined -O3 build passes every pose parity test
// (the one recorded failure was the test's own: the Rodrigues round trip is not unique beyond |r| = pi) and is faster
// (24 456 tag poses 0.45 -> 0.31 ms, board pose 0.27 -> 0.25 ms; profiles/r02_e_pnp_inline.txt).
#include "pnp_core.h"

typedef double v4d __attribute__((ext_vector_type(4)));
#define TAG_STRIDE 66         // doubles per target in LDS: 8 x 8 + 2 (spreads the owners' writes over the banks)

__global__ __launch_bounds__(64) void k_pnp_tags_mfma(rcc_detection* __restrict__ det, const int32_t* __restrict__ ndet,
                                                      int nframes, int max_targets, double tag_size, int reference_mode, rcc_cam cam)
{
  __shared__ double buf[64 * TAG_STRIDE];
  const int lane = threadIdx.x;
  const int t = blockIdx.x * 64 + lane;
  const int f = t / max_targets, k = t - f * max_targets;
  const bool have = (t < nframes * max_targets) && (k < ndet[f < nframes ? f : 0]);
  rcc_detection* d = det + t;
  const double s2 = 0.5 * tag_size;
  double obj[12] = { -s2, -s2, 0, s2, -s2, 0, s2, s2, 0, -s2, s2, 0 };
  double img[8];
  for (int c = 0; c < 4; ++c) {
    double x = have ? d->corners[c][0] : 0.0, y = have ? d->corners[c][1] : 0.0;
    if (reference_mode) { x = (double)(int)x; y = (double)(int)y; }       // corner_detections.cpp:53-54
    img[2 * c] = x; img[2 * c + 1] = y;
  }
  rccpnp::Pts p{ obj, img, 4 };
  rccpnp::Cam cm;
  cm.fx = cam.fx; cm.fy = cam.fy; cm.cx = cam.cx; cm.cy = cam.cy;
  const bool has_dist = cam.model == RCC_DIST_PLUMB_BOB;
  for (int i = 0; i < 5; ++i) cm.k[i] = has_dist ? cam.D[i] : 0.0;
  cm.solver = cam.solver;
  double wsl[rccpnp::PNP_WS];
  const rccpnp::SerialPar par{ wsl };
  double prm[6] = { 0, 0, 0, 0, 0, 0 };
  int status = rccpnp::PNP_TOO_FEW;
  if (have) status = rccpnp::pose_init(par, p, cm, has_dist, prm);
  bool active = have && !(status == rccpnp::PNP_TOO_FEW || status == rccpnp::PNP_NONPLANAR);

  // zero the staging area once: a lane that never writes G leaves zeros, not stale LDS contents, to its pair partner's MFMA
  for (int i = lane; i < 64 * TAG_STRIDE; i += 64) buf[i] = 0.0;
  __syncthreads();

  // ---- CvLevMarq (pnp_core.h pose_lm), one lane per target, in lock-step rounds
  double* const A = wsl;              // 36
  double* const Ap = wsl + 36;        // 36
  double* const Lw = wsl + 72;        // 36
  double g[6], pprev[6], dl[6];
  int L = -3, it = 0;
  bool need = true;                   // this lane's next round starts with new normal equations
  double prevErr = 0.0;
  const double eps = 1.1920928955078125e-07;   // FLT_EPSILON
  double* const mine = buf + lane * TAG_STRIDE;
  const int mi = lane & 15, mk = lane >> 4;          // MFMA operand element of this lane: column mi of [G_a | G_b], row mk (+4)
  while (__any(active)) {
    if (active && need) {
      double R[9], dRdr[27];
      rccpnp::rodrigues_v2m(prm, R, dRdr);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        double uv[2], Ju[6], Jv[6];
        rccpnp::project_point(obj + 3 * i, R, dRdr, prm + 3, cm, uv, Ju, Jv);
        double* r0 = mine + (2 * i) * 8;
        double* r1 = r0 + 8;
#pragma unroll
        for (int a = 0; a < 6; ++a) { r0[a] = Ju[a]; r1[a] = Jv[a]; }
        r0[6] = uv[0] - img[2 * i]; r1[6] = uv[1] - img[2 * i + 1];
        r0[7] = 0.0; r1[7] = 0.0;
      }
    }
    __syncthreads();
    for (int pair = 0; pair < 32; ++pair) {
      const double* src = buf + (2 * pair + (mi >> 3)) * TAG_STRIDE + (mi & 7);
      const double x0 = src[mk * 8], x1 = src[(mk + 4) * 8];
      v4d acc = { 0.0, 0.0, 0.0, 0.0 };
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x1, acc, 0, 0, 0);
      // D[row = mk + 4 v][col = mi]: keep the two diagonal 8 x 8 blocks, over the pair's own (consumed) G
      double* dst = buf + (2 * pair + (mi >> 3)) * TAG_STRIDE + (mi & 7);
      if (mi < 8) { dst[mk * 8] = acc[0]; dst[(mk + 4) * 8] = acc[1]; }
      else        { dst[mk * 8] = acc[2]; dst[(mk + 4) * 8] = acc[3]; }
    }
    __syncthreads();
    if (active && need) {
#pragma unroll
      for (int a = 0; a < 6; ++a) {
#pragma unroll
        for (int b = 0; b < 6; ++b) A[a * 6 + b] = mine[a * 8 + b];
        g[a] = mine[a * 8 + 6];
      }
      const double S0 = mine[6 * 8 + 6];
#pragma unroll
      for (int a = 0; a < 6; ++a) pprev[a] = prm[a];
      if (it == 0) prevErr = sqrt(S0);
      need = false;
    }
    if (active) {
      const double p10[34] = { 1e-16, 1e-15, 1e-14, 1e-13, 1e-12, 1e-11, 1e-10, 1e-9, 1e-8, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1.0,
                               1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17 };
      const double lambda = p10[(L < -16 ? -16 : (L > 17 ? 17 : L)) + 16];
      for (int i = 0; i < 36; ++i) Ap[i] = A[i];
      for (int a = 0; a < 6; ++a) Ap[a * 6 + a] *= 1.0 + lambda;
      rccpnp::spd_solve<6>(cm.solver, Ap, g, dl, Lw);
      for (int a = 0; a < 6; ++a) prm[a] = pprev[a] - dl[a];
      const double errNorm = sqrt(rccpnp::pose_accumulate(par, prm, p, cm, (double*)nullptr, (double*)nullptr));
      if (errNorm > prevErr && ++L <= 16) {
        // rejected: same normal equations, more damping, next round
      } else {
        L = (L - 1 > -16) ? L - 1 : -16;
        double dn = 0.0, pn = 0.0;
        for (int a = 0; a < 6; ++a) { dn += (prm[a] - pprev[a]) * (prm[a] - pprev[a]); pn += pprev[a] * pprev[a]; }
        const double rel = sqrt(dn) / sqrt(pn);
        if (++it >= 20 || rel < eps) active = false;
        else { prevErr = errNorm; need = true; }
      }
    }
    __syncthreads();      // the next round's G writes must not overtake this round's reads of the staging area
  }
  if (!have) return;
  double e = 0.0;
  if (!(status == rccpnp::PNP_TOO_FEW || status == rccpnp::PNP_NONPLANAR))
    e = sqrt(rccpnp::pose_accumulate(par, prm, p, cm, (double*)nullptr, (double*)nullptr) / 4.0);
  else
    for (int c = 0; c < 6; ++c) prm[c] = 0.0;
  for (int c = 0; c < 3; ++c) { d->rvec[c] = prm[c]; d->tvec[c] = prm[3 + c]; }
  d->rms = e; d->pnp_status = status; d->pnp_iters = it;
}

hipError_t rcc_launch_pnp_tags_mfma(rcc_handle* h, int nframes, rcc_cam cam, hipStream_t s)
{
  const rcc_config& c = h->cfg;
  const int total = nframes * c.max_targets;
  if (total <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_pnp_tags_mfma, dim3((total + 63) / 64), dim3(64), 0, s, h->d_det, h->d_ndet, nframes, c.max_targets,
                     c.tag_size, c.reference_mode, cam);
  return hipGetLastError();
}
