// pnp_core.h -- stage a7/a8 arithmetic: planar PnP in the solvePnP(ITERATIVE) form, and Rodrigues.
//
// Drop-in target: cv::solvePnP(obj_pts, img_pts, kcam_matrix, kdistCoeffs, rvec, tvec, false,
// CV_ITERATIVE) at real_preprocessing/src/camera_pose.cpp:163 (points bl,br,tr,tl :152-155, object
// points (+-size/2, +-size/2, 0) :158-161, K/D as loaded at :59-64) and cv::Rodrigues at :93,:116,
// :164 and opt_visualization.cpp:36.  Algorithm: SURVEY.md appendix A.2-A.8 (OpenCV 3.4.x as
// published; not present in this image).
//
// Written as per-thread code with NO per-point storage: normal equations are accumulated point by
// point (JtJ 6x6, Jte 6; for the homography refinement 8x8, 8), normalised points are recomputed
// where needed.  One lane per target (k_pnp.hip) runs this as is; the wave-per-target kernel
// shares the small dense algebra and spreads the per-point work over lanes.
//
// RCC_HD lets a host build of the same functions be exercised on the CPU (tests/test_pnp_core_host).
#pragma once
#include <math.h>
#include <float.h>
#include <stdint.h>

#ifndef RCC_HD
#ifdef __HIPCC__
#define RCC_HD __host__ __device__
#else
#define RCC_HD
#endif
#endif

// RCC_PNP_NOINLINE (off by default) keeps the big routines out of line in device builds: an A/B knob.  Inlined is the
// measured-faster form and matches the oracle in every parity test (the miscompile suspected in
// round 1 was not reproducible).
#if defined(__HIPCC__) && defined(RCC_PNP_NOINLINE)
#define RCC_NI __attribute__((noinline))
#else
#define RCC_NI
#endif
// the whole solve is inlined into every kernel that calls it (device builds): left to the compiler, the wave-per-target instantiation
// -- three call sites -- stayed a FUNCTION: its arguments and results went through scratch memory, its prologue saved the caller's
// registers there, and the lattice + pose kernel was the only kernel of the step with a scratch allocation
#if defined(__HIP_DEVICE_COMPILE__)
#define RCC_SOLVE_INLINE __attribute__((always_inline)) inline
#else
#define RCC_SOLVE_INLINE inline
#endif

#ifndef RCC_DIST_PLUMB_BOB
#define RCC_DIST_NONE 0
#define RCC_DIST_PLUMB_BOB 1
#define RCC_DIST_FISHEYE 2
#endif

#ifndef RCC_PNP_PHASE
#define RCC_PNP_PHASE(k)      /* experiment builds: a time stamp per solver phase (k_pnp.hip) */
#endif
#ifndef RCC_PNP_TIC
#define RCC_PNP_TIC()         /* experiment builds: per-category time accumulators of the solver's inner steps */
#define RCC_PNP_TOC(cat)
#endif

namespace rccpnp {

enum { PNP_OK = 0, PNP_TOO_FEW = 1, PNP_NONPLANAR = 2, PNP_DEGENERATE = 3 };

struct Cam {
  double fx, fy, cx, cy;
  double k[5];  // k1,k2,p1,p2,k3 (zero when the model is NONE)
  int solver;   // 0: eigen-decomposition solves (as published), 1: Cholesky with eigen fallback
};

// ---- symmetric eigen-decomposition (cyclic Jacobi), n <= 9; V rows = eigenvectors, w descending
RCC_NI RCC_HD inline void jacobi_eigen_sym(int n, double* A, double* w, double* V)
{
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) V[i * n + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int p = 0; p < n; ++p) {
      diag += A[p * n + p] * A[p * n + p];
      for (int q = p + 1; q < n; ++q) off += A[p * n + q] * A[p * n + q];
    }
    if (off <= 1e-300 || off <= 1e-34 * diag) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        double apq = A[p * n + q];
        if (apq == 0.0) continue;
        double app = A[p * n + p], aqq = A[q * n + q];
        double theta = (aqq - app) / (2.0 * apq);
        double t = 1.0 / (fabs(theta) + sqrt(theta * theta + 1.0));
        if (theta < 0.0) t = -t;
        double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) {
          double akp = A[k * n + p], akq = A[k * n + q];
          A[k * n + p] = c * akp - s * akq;
          A[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {
          double apk = A[p * n + k], aqk = A[q * n + k];
          A[p * n + k] = c * apk - s * aqk;
          A[q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          double vpk = V[p * n + k], vqk = V[q * n + k];
          V[p * n + k] = c * vpk - s * vqk;
          V[q * n + k] = s * vpk + c * vqk;
        }
      }
  }
  for (int i = 0; i < n; ++i) w[i] = A[i * n + i];
  for (int i = 0; i < n - 1; ++i) {
    int m = i;
    for (int j = i + 1; j < n; ++j) if (w[j] > w[m]) m = j;
    if (m != i) {
      double t = w[i]; w[i] = w[m]; w[m] = t;
      for (int k = 0; k < n; ++k) { double u = V[i * n + k]; V[i * n + k] = V[m * n + k]; V[m * n + k] = u; }
    }
  }
}

// 3x3 version with static indices (registers on the device); same sweep rule, same ordering
RCC_HD inline void jacobi_eigen_sym3(double* A, double* w, double* V)
{
#pragma unroll
  for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    const double diag = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
    if (off <= 1e-300 || off <= 1e-34 * diag) break;
#pragma unroll
    for (int pq = 0; pq < 3; ++pq) {
      const int p = (pq == 2) ? 1 : 0, q = (pq == 0) ? 1 : 2;
      const double apq = A[p * 3 + q];
      if (apq == 0.0) continue;
      const double app = A[p * 3 + p], aqq = A[q * 3 + q];
      const double theta = (aqq - app) / (2.0 * apq);
      double t = 1.0 / (fabs(theta) + sqrt(theta * theta + 1.0));
      if (theta < 0.0) t = -t;
      const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double akp = A[k * 3 + p], akq = A[k * 3 + q];
        A[k * 3 + p] = c * akp - sn * akq;
        A[k * 3 + q] = sn * akp + c * akq;
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double apk = A[p * 3 + k], aqk = A[q * 3 + k];
        A[p * 3 + k] = c * apk - sn * aqk;
        A[q * 3 + k] = sn * apk + c * aqk;
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double vpk = V[p * 3 + k], vqk = V[q * 3 + k];
        V[p * 3 + k] = c * vpk - sn * vqk;
        V[q * 3 + k] = sn * vpk + c * vqk;
      }
    }
  }
  w[0] = A[0]; w[1] = A[4]; w[2] = A[8];
  // sort descending with static compare-exchanges (0,1) (0,2) (1,2); rows of V follow
#pragma unroll
  for (int s3 = 0; s3 < 3; ++s3) {
    const int i = (s3 == 2) ? 1 : 0, j = (s3 == 0) ? 1 : 2;
    if (w[j] > w[i]) {
      double t = w[i]; w[i] = w[j]; w[j] = t;
#pragma unroll
      for (int k = 0; k < 3; ++k) { double u = V[i * 3 + k]; V[i * 3 + k] = V[j * 3 + k]; V[j * 3 + k] = u; }
    }
  }
}

// x = pinv(A) b, symmetric A (n <= 8), singular directions dropped as cv::solve(DECOMP_SVD) does.
// T, V: n*n scratch; w: n scratch.
RCC_NI RCC_HD inline void sym_solve(int n, const double* A, const double* b, double* x, double* T, double* V, double* w)
{
  for (int i = 0; i < n * n; ++i) T[i] = A[i];
  jacobi_eigen_sym(n, T, w, V);
  double thr = 0.0;
  for (int i = 0; i < n; ++i) thr += fabs(w[i]);
  thr *= 2.0 * DBL_EPSILON;
  for (int k = 0; k < n; ++k) x[k] = 0.0;
  for (int i = 0; i < n; ++i) {
    if (fabs(w[i]) <= thr) continue;
    double s = 0.0;
    for (int k = 0; k < n; ++k) s += V[i * n + k] * b[k];
    s /= w[i];
    for (int k = 0; k < n; ++k) x[k] += s * V[i * n + k];
  }
}

// Cholesky solve for the symmetric positive definite normal equations (N = 6 or 8, compile time:
// fully unrolled, the factor lives in registers).  The published
// algorithm solves them by SVD (A.8) / eigen-decomposition (A.4); for a well-conditioned SPD matrix
// the solutions agree to ~1e-12 relative.  A pivot below 1e-13 of the largest diagonal entry means
// the SVD path would have dropped a direction: then fall back to sym_solve, which does.
// The matrix is read ONCE (its lower triangle, one batch of loads) and the factor lives in registers (static indices, fully
// unrolled): with the wave-per-target mapping A sits in the LDS workspace, and a factor kept there made every one of
// its ~N^3/3 uses an LDS round trip on the solver's single dependency chain -- the 8x8 solve of the homography
// refinement took 6 us, the largest single piece of the board pose.  Damping: the diagonal is taken as A_ii + dadd[i]
// (dadd != null: the LMSolver form) or A_ii * dmul (CvLevMarq's), so the caller no longer builds a damped copy in memory.
// Same operations in the same order as the in-memory form: identical bits.
template <int N>
RCC_HD inline void spd_solve_damped(int solver, const double* A, const double* dadd, double dmul, const double* b, double* x)
{
  constexpr int T = N * (N + 1) / 2;
  double M[T], Lr[T];                      // lower triangles, row-major: (i, j <= i) at i (i + 1) / 2 + j
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) M[i * (i + 1) / 2 + j] = A[i * N + j];
#pragma unroll
  for (int i = 0; i < N; ++i) M[i * (i + 1) / 2 + i] = dadd ? M[i * (i + 1) / 2 + i] + dadd[i] : M[i * (i + 1) / 2 + i] * dmul;
  bool ok = (solver != 0);
  if (ok) {
    double dmax = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) if (M[i * (i + 1) / 2 + i] > dmax) dmax = M[i * (i + 1) / 2 + i];
    const double tiny = 1e-13 * dmax;
    ok = dmax > 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      double d = M[j * (j + 1) / 2 + j];
#pragma unroll
      for (int k = 0; k < j; ++k) d -= Lr[j * (j + 1) / 2 + k] * Lr[j * (j + 1) / 2 + k];
      if (!(d > tiny)) ok = false;
      // the diagonal holds 1 / L_jj: the substitutions below multiply (an fp64 division or square root is a
      // 10-20 instruction sequence on the solver's single dependency chain)
      const double idj = 1.0 / sqrt(d > tiny ? d : 1.0);
      Lr[j * (j + 1) / 2 + j] = idj;
#pragma unroll
      for (int i = j + 1; i < N; ++i) {
        double t = M[i * (i + 1) / 2 + j];
#pragma unroll
        for (int k = 0; k < j; ++k) t -= Lr[i * (i + 1) / 2 + k] * Lr[j * (j + 1) / 2 + k];
        Lr[i * (i + 1) / 2 + j] = t * idj;
      }
    }
  }
  if (!ok) {
    // eigen-decomposition path (as published; also the ill-conditioned fallback)
    double A2[N * N], b2[N], x2[N], Tm[N * N], V[N * N], w[N];
    for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) A2[i * N + j] = (j <= i) ? M[i * (i + 1) / 2 + j] : M[j * (j + 1) / 2 + i];
    for (int i = 0; i < N; ++i) b2[i] = b[i];
    sym_solve(N, A2, b2, x2, Tm, V, w);
    for (int i = 0; i < N; ++i) x[i] = x2[i];
    return;
  }
  double y[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {            // L y = b
    double t = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) t -= Lr[i * (i + 1) / 2 + k] * y[k];
    y[i] = t * Lr[i * (i + 1) / 2 + i];
  }
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {       // L^T x = y
    double t = y[i];
#pragma unroll
    for (int k = i + 1; k < N; ++k) t -= Lr[k * (k + 1) / 2 + i] * x[k];
    x[i] = t * Lr[i * (i + 1) / 2 + i];
  }
}
template <int N>
RCC_HD inline void spd_solve(int solver, const double* A, const double* b, double* x, double* /* workspace of the in-memory form: unused */)
{
  spd_solve_damped<N>(solver, A, (const double*)nullptr, 1.0, b, x);
}

// max_a (A^-1)[a][a] of a symmetric positive definite N x N matrix through its Cholesky factor: A^-1 = L^-T L^-1, so
// the a-th diagonal entry is the squared norm of column a of L^-1 (forward substitution on e_a).  Returns 0 when a
// pivot says the matrix is numerically rank deficient (the caller then takes the eigen-decomposition, which drops
// those directions as the published algorithm does).
template <int N>
RCC_HD inline int inv_diag_max_spd(const double* A, double* /* workspace of the in-memory form: unused */, double* maxdiag)
{
  constexpr int T = N * (N + 1) / 2;
  double M[T], Lr[T];                      // lower triangles in registers (see spd_solve_damped)
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) M[i * (i + 1) / 2 + j] = A[i * N + j];
  double dmax = 0.0;
#pragma unroll
  for (int i = 0; i < N; ++i) if (M[i * (i + 1) / 2 + i] > dmax) dmax = M[i * (i + 1) / 2 + i];
  if (!(dmax > 0.0)) return 0;
  const double tiny = 1e-13 * dmax;
  bool ok = true;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    double d = M[j * (j + 1) / 2 + j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= Lr[j * (j + 1) / 2 + k] * Lr[j * (j + 1) / 2 + k];
    if (!(d > tiny)) ok = false;
    const double idj = 1.0 / sqrt(d > tiny ? d : 1.0);
    Lr[j * (j + 1) / 2 + j] = idj;                       // 1 / L_jj
#pragma unroll
    for (int i = j + 1; i < N; ++i) {
      double t = M[i * (i + 1) / 2 + j];
#pragma unroll
      for (int k = 0; k < j; ++k) t -= Lr[i * (i + 1) / 2 + k] * Lr[j * (j + 1) / 2 + k];
      Lr[i * (i + 1) / 2 + j] = t * idj;
    }
  }
  if (!ok) return 0;
  double best = 0.0;
#pragma unroll
  for (int a = 0; a < N; ++a) {
    double y[N];
    y[a] = Lr[a * (a + 1) / 2 + a];
    double s = y[a] * y[a];
#pragma unroll
    for (int k = a + 1; k < N; ++k) {
      double t = 0.0;
#pragma unroll
      for (int j = a; j < k; ++j) t -= Lr[k * (k + 1) / 2 + j] * y[j];
      y[k] = t * Lr[k * (k + 1) / 2 + k];
      s += y[k] * y[k];
    }
    if (s > best) best = s;
  }
  *maxdiag = best;
  return 1;
}

// Eigenvector of the smallest eigenvalue of a symmetric positive semi-definite n x n matrix
// (n <= 9): shifted inverse iteration on a Cholesky factor of M + delta*I.  The DLT matrix L^T L
// has one eigenvalue that is (numerically) zero for consistent correspondences and a gap of many
// orders of magnitude above it, so three iterations reach rounding level.  Returns 0 if the
// factorization breaks down (caller then uses the Jacobi decomposition).  The published algorithm
// takes this vector from a full eigen-decomposition (A.4); the vector is the same up to sign, and
// the homography is normalised by H[2][2] afterwards.
template <int N>
RCC_HD inline int smallest_eigvec_psd(const double* Min, double* x /* N */, double* /* workspace of the in-memory form: unused */)
{
  constexpr int T = N * (N + 1) / 2;
  double M[T], Lr[T];                      // lower triangles in registers (see spd_solve_damped)
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) M[i * (i + 1) / 2 + j] = Min[i * N + j];
  double tr = 0.0;
#pragma unroll
  for (int i = 0; i < N; ++i) tr += M[i * (i + 1) / 2 + i];
  if (!(tr > 0.0)) return 0;
  const double delta = 1e-14 * tr;
  bool ok = true;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    double d = M[j * (j + 1) / 2 + j] + delta;
#pragma unroll
    for (int k = 0; k < j; ++k) d -= Lr[j * (j + 1) / 2 + k] * Lr[j * (j + 1) / 2 + k];
    if (!(d > 0.0)) ok = false;
    const double idj = 1.0 / sqrt(d > 0.0 ? d : 1.0);
    Lr[j * (j + 1) / 2 + j] = idj;                       // the diagonal holds 1 / L_jj (see spd_solve_damped)
#pragma unroll
    for (int i = j + 1; i < N; ++i) {
      double t = M[i * (i + 1) / 2 + j];
#pragma unroll
      for (int k = 0; k < j; ++k) t -= Lr[i * (i + 1) / 2 + k] * Lr[j * (j + 1) / 2 + k];
      Lr[i * (i + 1) / 2 + j] = t * idj;
    }
  }
  if (!ok) return 0;
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] = 1.0 - 0.07 * i;
#pragma unroll 1
  for (int it = 0; it < 4; ++it) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      double t = x[i];
#pragma unroll
      for (int k = 0; k < i; ++k) t -= Lr[i * (i + 1) / 2 + k] * x[k];
      x[i] = t * Lr[i * (i + 1) / 2 + i];
    }
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
      double t = x[i];
#pragma unroll
      for (int k = i + 1; k < N; ++k) t -= Lr[k * (k + 1) / 2 + i] * x[k];
      x[i] = t * Lr[i * (i + 1) / 2 + i];
    }
    double nr = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) nr += x[i] * x[i];
    if (!(nr > 0.0) || !isfinite(nr)) return 0;
    nr = 1.0 / sqrt(nr);
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] *= nr;
  }
  return 1;
}

// How the per-point loops are spread: serially in one thread, or over the 64 lanes of a wavefront
// (each lane takes points lane, lane+64, ...; sums are combined by an xor butterfly so that every
// lane ends up with the same value and the scalar algebra that follows stays wave-uniform).
// Both carry `w`: a workspace of PNP_WS doubles for the wave-uniform matrices (normal equations, Cholesky
// factors).  One copy per solver instance: per thread for SerialPar, per wavefront (in LDS) for WavePar --
// replicated in the registers of 64 lanes these matrices spill to scratch, and every spilled access costs a
// memory round trip on the solver's single dependency chain.
enum { PNP_WS = 320 };   // [0,192) matrices and factors, [192,256) reduction totals, [256,320) normalised points (<= 64)
struct SerialPar {
  double* w;
  static constexpr bool kGram = false;       // no matrix-core accumulation for a single thread
  RCC_HD bool gram_ok(int) const { return false; }
  template <int NC, bool BORDER> RCC_HD void gram(const double*, const double*, int, double*, int, double*) const {}
  RCC_HD double* ws() const { return w; }
  RCC_HD int first() const { return 0; }
  RCC_HD int step() const { return 1; }
  RCC_HD double sum(double v) const { return v; }
  RCC_HD double max(double v) const { return v; }
  // out[q] = sum over the solver's lanes of v[q], q < N (here: one lane)
  template <int N> RCC_HD void reduce_store(const double* v, double* out) const { for (int q = 0; q < N; ++q) out[q] = v[q]; }
  // full symmetric N x N matrix from its packed upper triangle (row-major: (0,0) (0,1) .. (0,N-1) (1,1) ..)
  template <int N> RCC_HD void unpack_sym(const double* tri, double* A) const
  {
    for (int r = 0, k = 0; r < N; ++r)
      for (int c = r; c < N; ++c, ++k) { A[r * N + c] = tri[k]; A[c * N + r] = tri[k]; }
  }
};
#ifdef __HIPCC__
}  // namespace rccpnp
#include "wave_reduce.h"
namespace rccpnp {
struct WavePar {
  int lane;
  unsigned ws_lds;   // LDS byte offset of the wavefront's workspace.  ws() rebuilds the pointer from it with an explicit
                     // address-space cast, so that inside the out-of-line solver routines the accesses compile to ds_*
                     // instructions: as a plain generic pointer they were FLAT operations, which go through the CU's
                     // vector-memory address path (16 cycles per wave instruction, shared by the four wavefronts of a CU)
  unsigned g_lds;    // LDS byte offset of a GRAM_ROWS x GRAM_STRIDE-double staging area for the matrix-core accumulation (offset 0 is a valid one); GRAM_NONE: no area
  __device__ double* ws() const { return (double*)(__attribute__((address_space(3))) double*)(size_t)ws_lds; }
  __device__ int first() const { return lane; }
  __device__ int step() const { return 64; }

  // ---- normal equations on the matrix cores (BASELINE.json north_star: "MFMA only for the small batched JtJ/Jtr
  // normal-equation blocks of the PnP Gauss-Newton").  Every lane owns one point (n <= 64) and hands over its two rows
  // g0, g1 of G = [J | e] (NC <= 10 columns); G^T G holds J^T J, J^T e and |e|^2.  v_mfma_f64_16x16x4_f64 computes a
  // 16 x 16 product of depth 4 whose A operand (16 x 4: A[l & 15][l >> 4]) and B operand (4 x 16: B[l >> 4][l & 15]) are, for
  // G^T G, the SAME register: lane l supplies G[4 s + (l >> 4)][l & 15] in step s (columns >= NC: zero).  2 n rows = n / 2
  // instructions chained through one accumulator (24 for the 48-corner board) replace the 45 (28) products per lane and
  // their 6-level recursive-halving reduction (~450 / 300 vector instructions on the solver's single chain).  The rows
  // cross the lanes through an LDS staging area: 18 writes per lane, one contiguous 512-byte read per step.
  // Result D[4 v + (l >> 4)][l & 15] in acc[v]: BORDER: the leading (NC-1) x (NC-1) block goes to dst (row stride dstride)
  // and column NC-1 (J^T e, and |e|^2 last) to border[0..NC-1]; else the whole NC x NC matrix to dst.
  enum { GRAM_ROWS = 128, GRAM_STRIDE = 10 };
  static constexpr unsigned GRAM_NONE = 0xFFFFFFFFu;
  static constexpr bool kGram = true;
  // MEASURED, NOT KEPT: compiled in only with -DRCC_PNP_GRAM.  On the 48-corner board the pose got
  // 18 us per frame SLOWER (95 -> 113 us median, scratch/t_grid_trace.py): gfx950 runs this instruction at 16 passes, so the
  // 24 chained steps hold the wave for ~1500 cycles where the 180 vector instructions + 48 exchange steps they replace
  // take about as many issue slots, and the rows' trip through LDS comes on top.
#ifdef RCC_PNP_GRAM
  __device__ bool gram_ok(int n) const { return g_lds != GRAM_NONE && n <= 64; }
#else
  __device__ bool gram_ok(int) const { return false; }
#endif
  template <int NC, bool BORDER>
  __device__ __forceinline__ void gram(const double* g0, const double* g1, int nrows, double* dst, int dstride, double* border) const
  {
    static_assert(NC <= GRAM_STRIDE, "staging row too short");
    typedef double v4d_t __attribute__((ext_vector_type(4)));
    double* const G = (double*)(__attribute__((address_space(3))) double*)(size_t)g_lds;
    double* const r0 = G + (2 * lane) * GRAM_STRIDE;
#pragma unroll
    for (int c = 0; c < NC; ++c) { r0[c] = g0[c]; r0[GRAM_STRIDE + c] = g1[c]; }
    __syncthreads();
    const int mi = lane & 15, mk = lane >> 4;
    const double* const src = G + mk * GRAM_STRIDE + (mi < NC ? mi : 0);
    v4d_t acc = { 0.0, 0.0, 0.0, 0.0 };
    const int steps = (nrows + 3) >> 2;                        // wave-uniform; rows beyond nrows were written as zeros
    for (int s0 = 0; s0 < steps; s0 += 8) {
      double x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = (s0 + u < steps && mi < NC) ? src[(4 * (s0 + u)) * GRAM_STRIDE] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u], x[u], acc, 0, 0, 0);
    }
    // acc[v] = D[4 v + mk][mi]
    if (mi < NC) {
#pragma unroll
      for (int v = 0; v < 3; ++v) {
        const int row = 4 * v + mk;
        if (row < NC) {
          const double val = (v == 0) ? acc[0] : (v == 1) ? acc[1] : acc[2];
          if (BORDER) {
            if (mi < NC - 1 && row < NC - 1) dst[row * dstride + mi] = val;
            if (mi == NC - 1) border[row] = val;
          } else {
            dst[row * dstride + mi] = val;
          }
        }
      }
    }
    __syncthreads();
  }
  __device__ double sum(double v) const { return wred::all_sum(lane, v); }
  __device__ double max(double v) const {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { double o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
    return v;
  }

  // N sums over the 64 lanes at once (N <= 64), by recursive halving (wave_reduce.h); the total of quantity q
  // ends in lane q * (64 / P), which stores it
  template <int N>
  __device__ __forceinline__ void reduce_store(const double* vin, double* out) const
  {
    constexpr int P = N > 32 ? 64 : N > 16 ? 32 : N > 8 ? 16 : N > 4 ? 8 : N > 2 ? 4 : N > 1 ? 2 : 1;
    double v[P];
#pragma unroll
    for (int q = 0; q < P; ++q) v[q] = q < N ? vin[q] : 0.0;
    wred::reduce_scatter<P>(lane, v);
    constexpr int G = 64 / P;
    const int q = lane / G;
    if ((lane % G) == 0 && q < N) out[q] = v[0];
  }
  template <int N>
  __device__ __forceinline__ void unpack_sym(const double* tri, double* A) const
  {
#pragma unroll
    for (int base = 0; base < N * N; base += 64) {
      const int l = base + lane;
      if (l < N * N) {
        const int r = l / N, c = l - r * N;
        const int a = r < c ? r : c, b = r < c ? c : r;
        A[l] = tri[a * N - (a * (a - 1)) / 2 + (b - a)];
      }
    }
  }
};
#endif

RCC_HD inline void mat3_mul(const double* A, const double* B, double* C)
{
  double T[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) T[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
#pragma unroll
  for (int i = 0; i < 9; ++i) C[i] = T[i];
}

RCC_HD inline double mat3_det(const double* M)
{
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// ---- a8 Rodrigues (appendix A.6) ---------------------------------------------------------------
RCC_HD inline void rodrigues_v2m(const double r[3], double R[9], double* J /* 27 or null */)
{
  double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  if (theta < DBL_EPSILON) {
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    if (J) {
#pragma unroll
      for (int i = 0; i < 27; ++i) J[i] = 0.0;
      J[5] = J[15] = J[19] = -1.0;
      J[7] = J[11] = J[21] = 1.0;
    }
    return;
  }
  double c, s;
#if defined(__HIP_DEVICE_COMPILE__)
  sincos(theta, &s, &c);          // one range reduction for both (the same values as sin() and cos())
#else
  c = cos(theta); s = sin(theta);
#endif
  double c1 = 1.0 - c, it = 1.0 / theta;
  double rx = r[0] * it, ry = r[1] * it, rz = r[2] * it;
  double rrt[9] = { rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz };
  double rxm[9] = { 0, -rz, ry, rz, 0, -rx, -ry, rx, 0 };
#pragma unroll
  for (int k = 0; k < 9; ++k) R[k] = c * ((k % 4 == 0) ? 1.0 : 0.0) + c1 * rrt[k] + s * rxm[k];
  if (J) {
    double drrt[27] = { rx + rx, ry, rz, ry, 0, 0, rz, 0, 0,
                        0, rx, 0, rx, ry + ry, rz, 0, rz, 0,
                        0, 0, rx, 0, 0, ry, rx, ry, rz + rz };
    const double drxm[27] = { 0, 0, 0, 0, 0, -1, 0, 1, 0,
                              0, 0, 1, 0, 0, 0, -1, 0, 0,
                              0, -1, 0, 1, 0, 0, 0, 0, 0 };
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      double ri = (i == 0) ? rx : (i == 1) ? ry : rz;
      double a0 = -s * ri, a1 = (s - 2.0 * c1 * it) * ri, a2 = c1 * it, a3 = (c - s * it) * ri, a4 = s * it;
#pragma unroll
      for (int k = 0; k < 9; ++k)
        J[i * 9 + k] = a0 * ((k % 4 == 0) ? 1.0 : 0.0) + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * rxm[k] + a4 * drxm[i * 9 + k];
    }
  }
}

RCC_HD inline void orthonormalise3(const double* M, double* Q)
{
  double MtM[9], w[3], V[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) MtM[i * 3 + j] = M[i] * M[j] + M[3 + i] * M[3 + j] + M[6 + i] * M[6 + j];
  // A matrix that already is a rotation to rounding (a product of rotations, as at the end of the initial pose): the
  // nearest rotation is M (M^T M)^(-1/2) = M (3 I - M^T M) / 2 + O(delta^2), delta = |M^T M - I| <= 1e-12 -- exact to the
  // last bit that matters, without the eigen-decomposition (3 sweeps of 3 rotations, each a division and two square
  // roots on the solver's single chain: ~6 us of the board pose).
  double dev = 0.0;
#pragma unroll
  for (int i = 0; i < 9; ++i) { const double e = fabs(MtM[i] - ((i % 4 == 0) ? 1.0 : 0.0)); dev = e > dev ? e : dev; }
  if (dev <= 1e-12) {
    double S1[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) S1[i] = ((i % 4 == 0) ? 1.5 : 0.0) - 0.5 * MtM[i];
    mat3_mul(M, S1, Q);
    return;
  }
  jacobi_eigen_sym3(MtM, w, V);
  double S[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double a = 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        double iw = (w[k] > 1e-300) ? 1.0 / sqrt(w[k]) : 0.0;
        a += V[k * 3 + i] * iw * V[k * 3 + j];
      }
      S[i * 3 + j] = a;
    }
  mat3_mul(M, S, Q);
}

RCC_HD inline void rodrigues_m2v(const double Rin[9], double r[3])
{
  double R[9];
  orthonormalise3(Rin, R);
  double vx = R[7] - R[5], vy = R[2] - R[6], vz = R[3] - R[1];
  double s = sqrt((vx * vx + vy * vy + vz * vz) * 0.25);
  double c = (R[0] + R[4] + R[8] - 1.0) * 0.5;
  c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
  double theta = acos(c);
  if (s < 1e-5) {
    if (c > 0) { r[0] = r[1] = r[2] = 0.0; return; }
    double t;
    t = (R[0] + 1.0) * 0.5; double x = sqrt(t > 0.0 ? t : 0.0);
    t = (R[4] + 1.0) * 0.5; double y = sqrt(t > 0.0 ? t : 0.0) * (R[1] < 0 ? -1.0 : 1.0);
    t = (R[8] + 1.0) * 0.5; double z = sqrt(t > 0.0 ? t : 0.0) * (R[2] < 0 ? -1.0 : 1.0);
    if (fabs(x) < fabs(y) && fabs(x) < fabs(z) && ((R[5] > 0) != (y * z > 0))) z = -z;
    double nrm = sqrt(x * x + y * y + z * z);
    double k = theta / nrm;
    r[0] = x * k; r[1] = y * k; r[2] = z * k;
    return;
  }
  double vv = (1.0 / (2.0 * s)) * theta;
  r[0] = vx * vv; r[1] = vy * vv; r[2] = vz * vv;
}

// ---- A.2 one point through undistortPoints (5 fixed iterations) -----------------------------------
RCC_HD inline void undistort_point(const Cam& cm, bool has_dist, double u, double v, double& xo, double& yo)
{
  double x0 = (u - cm.cx) / cm.fx, y0 = (v - cm.cy) / cm.fy;
  double x = x0, y = y0;
  if (has_dist) {
    const double k1 = cm.k[0], k2 = cm.k[1], p1 = cm.k[2], p2 = cm.k[3], k3 = cm.k[4];
    for (int it = 0; it < 5; ++it) {
      double r2 = x * x + y * y;
      double icdist = 1.0 / (1.0 + ((k3 * r2 + k2) * r2 + k1) * r2);
      double dx = 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x);
      double dy = p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y;
      x = (x0 - dx) * icdist;
      y = (y0 - dy) * icdist;
    }
  }
  xo = x; yo = y;
}

// ---- A.7 one point through projectPoints, with the two Jacobian rows (6 columns each) ------------
RCC_HD inline void project_point(const double M[3], const double R[9], const double* dRdr, const double t[3],
                                 const Cam& cm, double uv[2], double* Ju /* 6 or null */, double* Jv)
{
  const double* k = cm.k;
  double X = M[0], Y = M[1], Z = M[2];
  double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
  double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
  double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
  z = z ? 1.0 / z : 1.0;
  x *= z; y *= z;
  double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
  double a1 = 2.0 * x * y, a2 = r2 + 2.0 * x * x, a3 = r2 + 2.0 * y * y;
  double cdist = 1.0 + k[0] * r2 + k[1] * r4 + k[4] * r6;
  double xd = x * cdist + k[2] * a1 + k[3] * a2;
  double yd = y * cdist + k[2] * a3 + k[3] * a1;
  uv[0] = xd * cm.fx + cm.cx;
  uv[1] = yd * cm.fy + cm.cy;
  if (!Ju) return;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double dxj, dyj;
    if (j < 3) {
      const double* d = dRdr + j * 9;
      double dx0 = X * d[0] + Y * d[1] + Z * d[2];
      double dy0 = X * d[3] + Y * d[4] + Z * d[5];
      double dz0 = X * d[6] + Y * d[7] + Z * d[8];
      dxj = z * (dx0 - x * dz0);
      dyj = z * (dy0 - y * dz0);
    } else {
      int q = j - 3;
      dxj = (q == 0) ? z : (q == 2 ? -x * z : 0.0);
      dyj = (q == 1) ? z : (q == 2 ? -y * z : 0.0);
    }
    double dr2 = 2.0 * x * dxj + 2.0 * y * dyj;
    double dcd = k[0] * dr2 + 2.0 * k[1] * r2 * dr2 + 3.0 * k[4] * r4 * dr2;
    double da1 = 2.0 * (x * dyj + y * dxj);
    Ju[j] = cm.fx * (dxj * cdist + x * dcd + k[2] * da1 + k[3] * (dr2 + 4.0 * x * dxj));
    Jv[j] = cm.fy * (dyj * cdist + y * dcd + k[2] * (dr2 + 4.0 * y * dyj) + k[3] * da1);
  }
}

// source of points: obj (n x 3), img (n x 2), in any address space the caller can read
struct Pts {
  const double* obj;
  const double* img;
  int n;
  const float* nm;   // optional: the n image points already through undistortPoints and rounded to float (pose_init fills it)
};

// in-plane coordinates (float32-rounded, as findHomography converts its inputs) of point i
RCC_HD inline void plane_point(const Pts& p, int i, const double* Rt, const double* Tt, float& Mx, float& My)
{
  const double* M = p.obj + 3 * i;
  Mx = (float)(Rt[0] * M[0] + Rt[1] * M[1] + Rt[2] * M[2] + Tt[0]);
  My = (float)(Rt[3] * M[0] + Rt[4] * M[1] + Rt[5] * M[2] + Tt[1]);
}
RCC_HD inline void norm_point(const Pts& p, int i, const Cam& cm, bool has_dist, float& mx, float& my)
{
  if (p.nm) { mx = p.nm[2 * i]; my = p.nm[2 * i + 1]; return; }
  double x, y;
  undistort_point(cm, has_dist, p.img[2 * i], p.img[2 * i + 1], x, y);
  mx = (float)x; my = (float)y;
}

// residual sum S = |r|^2 of the homography h (8 params) and, when A != null, JtJ (8x8), Jtr (8),
// max |r|
template <class Par>
RCC_HD inline double homography_accumulate(const Par& par, const double* h, const Pts& p, const double* Rt, const double* Tt,
                                           const Cam& cm, bool has_dist, double* A, double* v, double* rinf)
{
  if constexpr (Par::kGram) {
    if (A && par.gram_ok(p.n)) {
      // one point per lane; its two rows of G = [J | r] go to the matrix cores (WavePar::gram)
      double ga[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 }, gb[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
      double ri = 0.0;
      const int i = par.first();
      if (i < p.n) {
        float Mxf, Myf, mxf, myf;
        plane_point(p, i, Rt, Tt, Mxf, Myf);
        norm_point(p, i, cm, has_dist, mxf, myf);
        const double Mx = Mxf, My = Myf;
        double ww = h[6] * Mx + h[7] * My + 1.0;
        ww = fabs(ww) > DBL_EPSILON ? 1.0 / ww : 0.0;
        const double xi = (h[0] * Mx + h[1] * My + h[2]) * ww;
        const double yi = (h[3] * Mx + h[4] * My + h[5]) * ww;
        const double e0 = xi - (double)mxf, e1 = yi - (double)myf;
        ri = fabs(e0) > fabs(e1) ? fabs(e0) : fabs(e1);
        ga[0] = Mx * ww; ga[1] = My * ww; ga[2] = ww; ga[6] = -Mx * ww * xi; ga[7] = -My * ww * xi; ga[8] = e0;
        gb[3] = Mx * ww; gb[4] = My * ww; gb[5] = ww; gb[6] = -Mx * ww * yi; gb[7] = -My * ww * yi; gb[8] = e1;
      }
      if (rinf) *rinf = par.max(ri);
      double* const red = par.ws() + 192;
      par.template gram<9, true>(ga, gb, 2 * p.n, A, 8, red + 36);      // A = JtJ, red[36..43] = Jtr, red[44] = S
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = red[36 + r];
      return red[44];
    }
  }
  // q[0..35]: this lane's share of JtJ (upper triangle, row-major), q[36..43]: of Jtr, q[44]: of S
  double q[45];
#pragma unroll
  for (int i = 0; i < 45; ++i) q[i] = 0.0;
  double ri = 0.0;
  for (int i = par.first(); i < p.n; i += par.step()) {
    float Mxf, Myf, mxf, myf;
    plane_point(p, i, Rt, Tt, Mxf, Myf);
    norm_point(p, i, cm, has_dist, mxf, myf);
    double Mx = Mxf, My = Myf;
    double ww = h[6] * Mx + h[7] * My + 1.0;
    ww = fabs(ww) > DBL_EPSILON ? 1.0 / ww : 0.0;
    double xi = (h[0] * Mx + h[1] * My + h[2]) * ww;
    double yi = (h[3] * Mx + h[4] * My + h[5]) * ww;
    double e0 = xi - (double)mxf, e1 = yi - (double)myf;
    q[44] += e0 * e0;
    q[44] += e1 * e1;
    if (fabs(e0) > ri) ri = fabs(e0);
    if (fabs(e1) > ri) ri = fabs(e1);
    if (A) {
      double a[8] = { Mx * ww, My * ww, ww, 0, 0, 0, -Mx * ww * xi, -My * ww * xi };
      double b[8] = { 0, 0, 0, Mx * ww, My * ww, ww, -Mx * ww * yi, -My * ww * yi };
#pragma unroll
      for (int r = 0, k = 0; r < 8; ++r) {
#pragma unroll
        for (int c = r; c < 8; ++c, ++k) q[k] += a[r] * a[c] + b[r] * b[c];
        q[36 + r] += a[r] * e0 + b[r] * e1;
      }
    }
  }
  if (rinf) *rinf = par.max(ri);
  double* const red = par.ws() + 192;         // 45 totals
  if (A) {
    par.template reduce_store<32>(q, red);            // 45 = 32 + 13: two halvings (31 + 17 exchange steps, 64 + 32 live
    par.template reduce_store<13>(q + 32, red + 32);  // registers) instead of one padded to 64 (63 steps, 128 registers)
    par.template unpack_sym<8>(red, A);
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = red[36 + r];
    return red[44];
  }
  par.template reduce_store<1>(q + 44, red + 44);      // the same pairing tree as above: identical rounding
  return red[44];
}

// A.4, N > 4: the LMSolver refinement used by findHomography(method 0), <= 10 iterations
template <class Par>
RCC_NI RCC_HD inline void homography_refine(const Par& par, double* h, const Pts& p, const double* Rt, const double* Tt, const Cam& cm, bool has_dist)
{
  const int P = 8, maxIters = 10;
  const double epsx = FLT_EPSILON, epsf = FLT_EPSILON;
  double x[8], xd[8], v[8], d[8], Dg[8], tmp[8];
  double* const A = par.ws();                 // 64
  double* const Lw = par.ws() + 128;          // 64: Cholesky factor
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = h[i];
  double rinf = 0.0;
  RCC_PNP_TIC();
  double S = homography_accumulate(par, x, p, Rt, Tt, cm, has_dist, A, v, &rinf);
  RCC_PNP_TOC(0);
#pragma unroll
  for (int i = 0; i < P; ++i) Dg[i] = A[i * P + i];
  const double Rlo = 0.25, Rhi = 0.75;
  double lambda = 1.0, lc = 0.75;
  int iter = 0;
  for (;;) {
    double dadd[8];
#pragma unroll
    for (int i = 0; i < P; ++i) dadd[i] = lambda * Dg[i];
    spd_solve_damped<8>(cm.solver, A, dadd, 1.0, v, d);          // (A + lambda diag(Dg)) d = v
    RCC_PNP_TOC(2);
#pragma unroll
    for (int i = 0; i < P; ++i) xd[i] = x[i] - d[i];
    double Sd = homography_accumulate(par, xd, p, Rt, Tt, cm, has_dist, (double*)nullptr, (double*)nullptr, (double*)nullptr);
    RCC_PNP_TOC(1);
    double dS = 0.0;
#pragma unroll
    for (int a = 0; a < P; ++a) {
      double s = 0.0;
#pragma unroll
      for (int b = 0; b < P; ++b) s += A[a * P + b] * d[b];
      tmp[a] = -s + 2.0 * v[a];
      dS += d[a] * tmp[a];
    }
    double Rr = (S - Sd) / (fabs(dS) > DBL_EPSILON ? dS : 1.0);
    if (Rr > Rhi) {
      lambda *= 0.5;
      if (lambda < lc) lambda = 0.0;
    } else if (Rr < Rlo) {
      double t = 0.0;
#pragma unroll
      for (int a = 0; a < P; ++a) t += d[a] * v[a];
      double nu = (Sd - S) / (fabs(t) > DBL_EPSILON ? t : 1.0) + 2.0;
      nu = nu < 2.0 ? 2.0 : (nu > 10.0 ? 10.0 : nu);
      if (lambda == 0.0) {
#ifdef RCC_PNP_TRACE_EIGEN
        RCC_PNP_TRACE_EIGEN(iter);
#endif
        // lc = 1 / max diagonal entry of pinv(JtJ) (LMSolver takes it from an SVD).  Every solve comes through here
        // once, near convergence; as a cyclic Jacobi decomposition of the 8x8 matrix it was three quarters of the
        // whole pose solve (~45 k of 60 k instructions).  For a positive definite JtJ the same number comes out of
        // a Cholesky factor; the decomposition remains for solver 0 and for a rank-deficient matrix.
        double maxval = DBL_EPSILON;
        double mdiag = 0.0;
        if (cm.solver == 1 && inv_diag_max_spd<8>(A, Lw, &mdiag)) {
          if (mdiag > maxval) maxval = mdiag;
        } else {
          double* const T = par.ws() + 64;    // Ap and the factor are dead here
          double* const V = par.ws() + 128;
          double w[8];
#pragma unroll
          for (int i = 0; i < 64; ++i) T[i] = A[i];
          jacobi_eigen_sym(P, T, w, V);
          double thr = 0.0;
#pragma unroll
          for (int i = 0; i < P; ++i) thr += fabs(w[i]);
          thr *= 2.0 * DBL_EPSILON;
#pragma unroll
          for (int a = 0; a < P; ++a) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < P; ++i) if (fabs(w[i]) > thr) s += V[i * P + a] * V[i * P + a] / w[i];
            if (fabs(s) > maxval) maxval = fabs(s);
          }
        }
        lambda = lc = 1.0 / maxval;
        nu *= 0.5;
      }
      lambda *= nu;
    }
    RCC_PNP_TOC(3);
    if (Sd < S) {
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = xd[i];
      S = homography_accumulate(par, x, p, Rt, Tt, cm, has_dist, A, v, &rinf);
      RCC_PNP_TOC(0);
    }
    ++iter;
    double dinf = 0.0;
#pragma unroll
    for (int i = 0; i < P; ++i) if (fabs(d[i]) > dinf) dinf = fabs(d[i]);
    if (!(iter < maxIters && dinf >= epsx && rinf >= epsf)) break;
  }
#ifdef RCC_PNP_TRACE_REFINE
  RCC_PNP_TRACE_REFINE(iter);
#endif
  RCC_PNP_PHASE(100 + iter);
#pragma unroll
  for (int i = 0; i < 8; ++i) h[i] = x[i];
}

// A.4: normalised DLT (+ refinement).  Returns 1 if H is finite.
template <class Par>
RCC_NI RCC_HD inline int find_homography(const Par& par, const Pts& p, const double* Rt, const double* Tt, const Cam& cm, bool has_dist, double H[9])
{
  const int n = p.n;
  double* const red4 = par.ws() + 192;
  double c4[4] = { 0, 0, 0, 0 };
  for (int i = par.first(); i < n; i += par.step()) {
    float Mx, My, mx, my;
    plane_point(p, i, Rt, Tt, Mx, My);
    norm_point(p, i, cm, has_dist, mx, my);
    c4[0] += Mx; c4[1] += My; c4[2] += mx; c4[3] += my;
  }
  par.template reduce_store<4>(c4, red4);
  double cMx = red4[0], cMy = red4[1], cmx = red4[2], cmy = red4[3];
  cMx /= n; cMy /= n; cmx /= n; cmy /= n;
  double s4[4] = { 0, 0, 0, 0 };
  for (int i = par.first(); i < n; i += par.step()) {
    float Mx, My, mx, my;
    plane_point(p, i, Rt, Tt, Mx, My);
    norm_point(p, i, cm, has_dist, mx, my);
    s4[0] += fabs(Mx - cMx); s4[1] += fabs(My - cMy);
    s4[2] += fabs(mx - cmx); s4[3] += fabs(my - cmy);
  }
  par.template reduce_store<4>(s4, red4 + 4);
  double sMx = red4[4], sMy = red4[5], smx = red4[6], smy = red4[7];
  if (fabs(sMx) < DBL_EPSILON || fabs(sMy) < DBL_EPSILON || fabs(smx) < DBL_EPSILON || fabs(smy) < DBL_EPSILON) return 0;
  smx = n / smx; smy = n / smy; sMx = n / sMx; sMy = n / sMy;
  double invHnorm[9] = { 1.0 / smx, 0, cmx, 0, 1.0 / smy, cmy, 0, 0, 1 };
  double Hnorm2[9] = { sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1 };
  double* const LtL = par.ws();               // 81
  bool on_mfma = false;
  if constexpr (Par::kGram) {
    if (par.gram_ok(n)) {
      double Lx[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 }, Ly[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
      const int i = par.first();
      if (i < n) {
        float Mxf, Myf, mxf, myf;
        plane_point(p, i, Rt, Tt, Mxf, Myf);
        norm_point(p, i, cm, has_dist, mxf, myf);
        const double x = (mxf - cmx) * smx, y = (myf - cmy) * smy;
        const double X = (Mxf - cMx) * sMx, Y = (Myf - cMy) * sMy;
        Lx[0] = X; Lx[1] = Y; Lx[2] = 1; Lx[6] = -x * X; Lx[7] = -x * Y; Lx[8] = -x;
        Ly[3] = X; Ly[4] = Y; Ly[5] = 1; Ly[6] = -y * X; Ly[7] = -y * Y; Ly[8] = -y;
      }
      par.template gram<9, false>(Lx, Ly, 2 * n, LtL, 9, (double*)nullptr);   // L^T L, the whole 9 x 9 matrix
      on_mfma = true;
    }
  }
  double ll[45];                              // this lane's share, upper triangle row-major
#pragma unroll
  for (int i = 0; i < 45; ++i) ll[i] = 0.0;
  if (!on_mfma) {
  for (int i = par.first(); i < n; i += par.step()) {
    float Mxf, Myf, mxf, myf;
    plane_point(p, i, Rt, Tt, Mxf, Myf);
    norm_point(p, i, cm, has_dist, mxf, myf);
    double x = (mxf - cmx) * smx, y = (myf - cmy) * smy;
    double X = (Mxf - cMx) * sMx, Y = (Myf - cMy) * sMy;
    double Lx[9] = { X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x };
    double Ly[9] = { 0, 0, 0, X, Y, 1, -y * X, -y * Y, -y };
#pragma unroll
    for (int j = 0, q = 0; j < 9; ++j)
#pragma unroll
      for (int k = j; k < 9; ++k, ++q) ll[q] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
  }
  par.template reduce_store<32>(ll, par.ws() + 192);
  par.template reduce_store<13>(ll + 32, par.ws() + 192 + 32);
  par.template unpack_sym<9>(par.ws() + 192, LtL);
  }
  double H0[9], T[9];
  if (!(cm.solver == 1 && smallest_eigvec_psd<9>(LtL, H0, par.ws() + 81))) {
    double* const L2 = LtL;                           // full decomposition (as published / fallback), in place
    double* const V = par.ws() + 81;
    double w[9];
    jacobi_eigen_sym(9, L2, w, V);
#pragma unroll
    for (int k = 0; k < 9; ++k) H0[k] = V[8 * 9 + k];
  }
  mat3_mul(invHnorm, H0, T);
  mat3_mul(T, Hnorm2, H);
  double s = 1.0 / H[8];
#pragma unroll
  for (int k = 0; k < 9; ++k) H[k] *= s;
  H[8] = 1.0;
  RCC_PNP_PHASE(9);
  if (n > 4) homography_refine(par, H, p, Rt, Tt, cm, has_dist);
  RCC_PNP_PHASE(10);
#pragma unroll
  for (int k = 0; k < 9; ++k) if (!isfinite(H[k])) return 0;
  return 1;
}

// normal equations of the pose LM at parameters prm (r,t): A = JtJ (6x6), g = Jte (6); returns |e|^2
template <class Par>
RCC_HD inline double pose_accumulate(const Par& par, const double* prm, const Pts& p, const Cam& cm, double* A, double* g)
{
  double R[9], dRdr[27];
  rodrigues_v2m(prm, R, A ? dRdr : nullptr);
  if constexpr (Par::kGram) {
    if (A && par.gram_ok(p.n)) {
      double gu[7] = { 0, 0, 0, 0, 0, 0, 0 }, gv[7] = { 0, 0, 0, 0, 0, 0, 0 };
      const int i = par.first();
      if (i < p.n) {
        double uv[2];
        project_point(p.obj + 3 * i, R, dRdr, prm + 3, cm, uv, gu, gv);      // the two Jacobian rows straight into G
        gu[6] = uv[0] - p.img[2 * i]; gv[6] = uv[1] - p.img[2 * i + 1];
      }
      double* const red = par.ws() + 192;
      par.template gram<7, true>(gu, gv, 2 * p.n, A, 6, red + 21);            // A = JtJ, red[21..26] = Jte, red[27] = S
#pragma unroll
      for (int a = 0; a < 6; ++a) g[a] = red[21 + a];
      return red[27];
    }
  }
  // q[0..20]: this lane's share of JtJ (upper triangle, row-major), q[21..26]: of Jte, q[27]: of S
  double q[28];
#pragma unroll
  for (int i = 0; i < 28; ++i) q[i] = 0.0;
  for (int i = par.first(); i < p.n; i += par.step()) {
    double uv[2], Ju[6], Jv[6];
    project_point(p.obj + 3 * i, R, dRdr, prm + 3, cm, uv, A ? Ju : nullptr, Jv);
    double e0 = uv[0] - p.img[2 * i], e1 = uv[1] - p.img[2 * i + 1];
    q[27] += e0 * e0;
    q[27] += e1 * e1;
    if (A) {
#pragma unroll
      for (int a = 0, k = 0; a < 6; ++a) {
#pragma unroll
        for (int b = a; b < 6; ++b, ++k) q[k] += Ju[a] * Ju[b] + Jv[a] * Jv[b];
        q[21 + a] += Ju[a] * e0 + Jv[a] * e1;
      }
    }
  }
  double* const red = par.ws() + 192;         // 28 totals
  if (A) {
    par.template reduce_store<28>(q, red);
    par.template unpack_sym<6>(red, A);
#pragma unroll
    for (int a = 0; a < 6; ++a) g[a] = red[21 + a];
    return red[27];
  }
  par.template reduce_store<1>(q + 27, red + 27);      // the same pairing tree as above: identical rounding
  return red[27];
}

// A.1, A.3, A.5: the initial pose.  Returns a PNP_* status; prm = (r, t).
template <class Par>
RCC_NI RCC_HD inline int pose_init(const Par& par, const Pts& p, const Cam& cm, bool has_dist, double prm[6])
{
  const int n = p.n;
#pragma unroll
  for (int i = 0; i < 6; ++i) prm[i] = 0.0;
  if (n < 4) return PNP_TOO_FEW;
  double Mc[3] = { 0, 0, 0 };
  for (int i = par.first(); i < n; i += par.step()) for (int k = 0; k < 3; ++k) Mc[k] += p.obj[3 * i + k];
#pragma unroll
  for (int k = 0; k < 3; ++k) Mc[k] = par.sum(Mc[k]) / n;
  double MM[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) MM[i] = 0.0;
  for (int i = par.first(); i < n; i += par.step()) {
    double d[3] = { p.obj[3 * i] - Mc[0], p.obj[3 * i + 1] - Mc[1], p.obj[3 * i + 2] - Mc[2] };
#pragma unroll
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) MM[a * 3 + b] += d[a] * d[b];
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) MM[i] = par.sum(MM[i]);
  double W[3], Vt[9];
  jacobi_eigen_sym3(MM, W, Vt);
  if (!(W[2] / W[1] < 1e-3)) return PNP_NONPLANAR;
  double Rt[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) Rt[k] = Vt[k];
  if (Vt[2] * Vt[2] + Vt[5] * Vt[5] < 1e-10) {
#pragma unroll
    for (int k = 0; k < 9; ++k) Rt[k] = (k % 4 == 0) ? 1.0 : 0.0;
  }
  if (mat3_det(Rt) < 0) for (int k = 0; k < 9; ++k) Rt[k] = -Rt[k];
  double Tt[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) Tt[a] = -(Rt[a * 3] * Mc[0] + Rt[a * 3 + 1] * Mc[1] + Rt[a * 3 + 2] * Mc[2]);
  double H[9], R[9], t[3], r[3];
  int status = PNP_OK;
  // the homography stages visit every image point many times: normalise each once (5 fixed-point iterations)
  Pts pc = p;
  if (n <= 64 && !p.nm) {
    float* nm = reinterpret_cast<float*>(par.ws() + 256);
    for (int i = par.first(); i < n; i += par.step()) norm_point(p, i, cm, has_dist, nm[2 * i], nm[2 * i + 1]);
    pc.nm = nm;
  }
  RCC_PNP_PHASE(8);
  if (find_homography(par, pc, Rt, Tt, cm, has_dist, H)) {
    RCC_PNP_PHASE(11);
    double h1[3] = { H[0], H[3], H[6] }, h2[3] = { H[1], H[4], H[7] };
    t[0] = H[2]; t[1] = H[5]; t[2] = H[8];
    double n1 = sqrt(h1[0] * h1[0] + h1[1] * h1[1] + h1[2] * h1[2]);
    double n2 = sqrt(h2[0] * h2[0] + h2[1] * h2[1] + h2[2] * h2[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) { h1[k] /= n1; h2[k] /= n2; t[k] *= 2.0 / (n1 + n2); }
    double h3[3] = { h1[1] * h2[2] - h1[2] * h2[1], h1[2] * h2[0] - h1[0] * h2[2], h1[0] * h2[1] - h1[1] * h2[0] };
    double R0[9] = { h1[0], h2[0], h3[0], h1[1], h2[1], h3[1], h1[2], h2[2], h3[2] };
    rodrigues_m2v(R0, r);
    rodrigues_v2m(r, R, nullptr);
    double t2[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) t2[a] = R[a * 3] * Tt[0] + R[a * 3 + 1] * Tt[1] + R[a * 3 + 2] * Tt[2] + t[a];
#pragma unroll
    for (int a = 0; a < 3; ++a) t[a] = t2[a];
    mat3_mul(R, Rt, R);
  } else {
    status = PNP_DEGENERATE;
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = (k % 4 == 0) ? 1.0 : 0.0;
    t[0] = t[1] = t[2] = 0.0;
  }
  rodrigues_m2v(R, r);
#pragma unroll
  for (int k = 0; k < 3; ++k) { prm[k] = r[k]; prm[3 + k] = t[k]; }
  return status;
}

// A.8: CvLevMarq, (J, err) form, 6 free parameters.  `accum(prm, A, g)` returns |e|^2 and fills
// A, g when A != null -- it is the only place the points are touched, so a caller can supply a
// wave-parallel (or MFMA) accumulation.
template <class Accum, class Par>
RCC_NI RCC_HD inline int pose_lm(double p[6], Accum accum, int solver, double* rms_sq_sum, const Par& par)
{
  double pprev[6], g[6], dl[6];
  double* const ws = par.ws();                // rebuilt here (not passed in) so that the address space is known: see WavePar
  double* const A = ws;                       // 36
  int L = -3, it = 0;
  double prevErr = 0.0;
  const int max_iter = 20;
  const double eps = FLT_EPSILON;
  RCC_PNP_TIC();
  for (;;) {
    double S0 = accum(p, A, g);
    RCC_PNP_TOC(4);
#pragma unroll
    for (int a = 0; a < 6; ++a) pprev[a] = p[a];
    if (it == 0) prevErr = sqrt(S0);
    double errNorm;
    for (;;) {
      // 10^L, L in [-16, 17]: exact decimal constants instead of exp(L log 10) (two libm calls per trial step)
      const double p10[34] = { 1e-16, 1e-15, 1e-14, 1e-13, 1e-12, 1e-11, 1e-10, 1e-9, 1e-8, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1.0,
                               1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17 };
      double lambda = p10[(L < -16 ? -16 : (L > 17 ? 17 : L)) + 16];
      spd_solve_damped<6>(solver, A, (const double*)nullptr, 1.0 + lambda, g, dl);     // diag * (1 + lambda)
      RCC_PNP_TOC(6);
#pragma unroll
      for (int a = 0; a < 6; ++a) p[a] = pprev[a] - dl[a];
      errNorm = sqrt(accum(p, (double*)nullptr, (double*)nullptr));
      RCC_PNP_TOC(5);
      if (errNorm > prevErr) {
        if (++L <= 16) continue;
      }
      break;
    }
    L = (L - 1 > -16) ? L - 1 : -16;
    double dn = 0.0, pn = 0.0;
#pragma unroll
    for (int a = 0; a < 6; ++a) { dn += (p[a] - pprev[a]) * (p[a] - pprev[a]); pn += pprev[a] * pprev[a]; }
    double rel = sqrt(dn) / sqrt(pn);
    if (++it >= max_iter || rel < eps) break;
    prevErr = errNorm;
  }
  if (rms_sq_sum) *rms_sq_sum = accum(p, (double*)nullptr, (double*)nullptr);
  return it;
}

template <class Par>
struct ParAccum {
  Par par;
  Pts p;
  Cam cm;
  RCC_HD double operator()(const double* prm, double* A, double* g) const { return pose_accumulate(par, prm, p, cm, A, g); }
};

// whole solve; Par = SerialPar: one thread per target; Par = WavePar: one wavefront per target
template <class Par>
RCC_HD RCC_SOLVE_INLINE int solve_pnp(const Par& par, const Pts& p, const Cam& cm_in, int dist_model, double rvec[3], double tvec[3], double* rms, int* iters)
{
  Cam cm = cm_in;
  const bool has_dist = (dist_model == RCC_DIST_PLUMB_BOB);
  if (!has_dist) for (int i = 0; i < 5; ++i) cm.k[i] = 0.0;
  double prm[6];
  if (rms) *rms = 0.0;
  if (iters) *iters = 0;
  int status = pose_init(par, p, cm, has_dist, prm);
  if (status == PNP_TOO_FEW || status == PNP_NONPLANAR) {
    for (int k = 0; k < 3; ++k) { rvec[k] = 0.0; tvec[k] = 0.0; }
    return status;
  }
  RCC_PNP_PHASE(12);
  ParAccum<Par> acc{ par, p, cm };
  double ss = 0.0;
  int it = pose_lm(prm, acc, cm.solver, &ss, par);
  for (int k = 0; k < 3; ++k) { rvec[k] = prm[k]; tvec[k] = prm[3 + k]; }
  if (rms) *rms = sqrt(ss / p.n);
  if (iters) *iters = it;
  return status;
}

}  // namespace rccpnp
