/*
 * orc_pnp.c -- CPU oracle, stages a7 (planar PnP, solvePnP ITERATIVE) and a8 (Rodrigues).
 * TEST INFRASTRUCTURE.
 *
 * Call site restated: real_preprocessing/src/camera_pose.cpp:163
 *     cv::solvePnP(obj_pts, img_pts, kcam_matrix, kdistCoeffs, rvec, tvec, false, CV_ITERATIVE)
 * with img_pts in the order bl,br,tr,tl (:152-155), obj_pts (+-size/2, +-size/2, 0) (:158-161),
 * K as 9 row-major doubles and D = (k1,k2,p1,p2,k3) (:59-64), then cv::Rodrigues (:164; also
 * :93,:116 and opt_visualization.cpp:36 for the matrix->vector direction).
 *
 * The arithmetic lives in OpenCV 3.4.4 (real_preprocessing/README.md:40), which is absent from
 * this image; what follows restates its published algorithm as SURVEY.md appendix A records it
 * ([U]: recollection of upstream, not checked against its source).  Steps: A.2 undistortPoints
 * (5 fixed iterations), A.3 planarity + in-plane frame, A.4 normalised-DLT homography on float32
 * points (+ LM refinement when N > 4), A.5 R,t from H with a Rodrigues round trip, A.6 Rodrigues,
 * A.7 projectPoints + Jacobian, A.8 CvLevMarq (lambda = 10^L, L0 = -3, diag*(1+lambda), SVD solve,
 * <= 20 iterations, eps = FLT_EPSILON).
 * Symmetric eigen-decomposition (cyclic Jacobi) stands in for the SVD of the symmetric matrices.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>
#include "orc.h"

/* ---- small dense helpers ------------------------------------------------------------------- */

/* cyclic Jacobi for a symmetric n x n matrix (n <= 9). w: eigenvalues, descending.
 * V: n x n, row i = eigenvector of w[i] (OpenCV cv::eigen convention). A is destroyed. */
void orc_jacobi_eigen_sym(int n, double* A, double* w, double* V)
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
        for (int k = 0; k < n; ++k) {       /* columns p,q of A */
          double akp = A[k * n + p], akq = A[k * n + q];
          A[k * n + p] = c * akp - s * akq;
          A[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {       /* rows p,q of A */
          double apk = A[p * n + k], aqk = A[q * n + k];
          A[p * n + k] = c * apk - s * aqk;
          A[q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {       /* accumulate vectors as rows */
          double vpk = V[p * n + k], vqk = V[q * n + k];
          V[p * n + k] = c * vpk - s * vqk;
          V[q * n + k] = s * vpk + c * vqk;
        }
      }
  }
  for (int i = 0; i < n; ++i) w[i] = A[i * n + i];
  for (int i = 0; i < n - 1; ++i) {      /* selection sort, descending */
    int m = i;
    for (int j = i + 1; j < n; ++j) if (w[j] > w[m]) m = j;
    if (m != i) {
      double t = w[i]; w[i] = w[m]; w[m] = t;
      for (int k = 0; k < n; ++k) { double u = V[i * n + k]; V[i * n + k] = V[m * n + k]; V[m * n + k] = u; }
    }
  }
}

/* x = pinv(A) b for symmetric A (n <= 8), by eigen-decomposition; singular directions
 * (|w| <= 2*DBL_EPSILON*sum|w|) are dropped, as cv::solve(DECOMP_SVD) / SVBkSb does. */
static void sym_solve(int n, const double* A, const double* b, double* x)
{
  double T[64], w[8], V[64];
  memcpy(T, A, sizeof(double) * n * n);
  orc_jacobi_eigen_sym(n, T, w, V);
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

static void mat3_mul(const double* A, const double* B, double* C)
{
  double T[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) T[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
  memcpy(C, T, sizeof(T));
}

static double mat3_det(const double* M)
{
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
         M[2] * (M[3] * M[7] - M[4] * M[6]);
}

/* ---- a8 Rodrigues (appendix A.6) -------------------------------------------------------------- */
void orc_rodrigues_v2m(const double r[3], double R[9], double J[27])
{
  double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  if (theta < DBL_EPSILON) {
    for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    if (J) {
      memset(J, 0, sizeof(double) * 27);
      J[5] = J[15] = J[19] = -1.0;
      J[7] = J[11] = J[21] = 1.0;
    }
    return;
  }
  double c = cos(theta), s = sin(theta), c1 = 1.0 - c, it = 1.0 / theta;
  double rx = r[0] * it, ry = r[1] * it, rz = r[2] * it;
  double rrt[9] = { rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz };
  double rxm[9] = { 0, -rz, ry, rz, 0, -rx, -ry, rx, 0 };
  for (int k = 0; k < 9; ++k) R[k] = c * ((k % 4 == 0) ? 1.0 : 0.0) + c1 * rrt[k] + s * rxm[k];
  if (J) {
    const double I9[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    double drrt[27] = { rx + rx, ry, rz, ry, 0, 0, rz, 0, 0,
                        0, rx, 0, rx, ry + ry, rz, 0, rz, 0,
                        0, 0, rx, 0, 0, ry, rx, ry, rz + rz };
    const double drxm[27] = { 0, 0, 0, 0, 0, -1, 0, 1, 0,
                              0, 0, 1, 0, 0, 0, -1, 0, 0,
                              0, -1, 0, 1, 0, 0, 0, 0, 0 };
    for (int i = 0; i < 3; ++i) {
      double ri = (i == 0) ? rx : (i == 1) ? ry : rz;
      double a0 = -s * ri, a1 = (s - 2.0 * c1 * it) * ri, a2 = c1 * it, a3 = (c - s * it) * ri, a4 = s * it;
      for (int k = 0; k < 9; ++k)
        J[i * 9 + k] = a0 * I9[k] + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * rxm[k] + a4 * drxm[i * 9 + k];
    }
  }
}

/* nearest rotation U*V^T of M via the symmetric eigen-decomposition of M^T M */
static void orthonormalise3(const double* M, double* Q)
{
  double MtM[9], w[3], V[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) MtM[i * 3 + j] = M[i] * M[j] + M[3 + i] * M[3 + j] + M[6 + i] * M[6 + j];
  orc_jacobi_eigen_sym(3, MtM, w, V);
  /* Q = M * V^T diag(1/sqrt(w)) V   (V rows = vectors) */
  double S[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double a = 0.0;
      for (int k = 0; k < 3; ++k) {
        double iw = (w[k] > 1e-300) ? 1.0 / sqrt(w[k]) : 0.0;
        a += V[k * 3 + i] * iw * V[k * 3 + j];
      }
      S[i * 3 + j] = a;
    }
  mat3_mul(M, S, Q);
}

void orc_rodrigues_m2v(const double Rin[9], double r[3])
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

/* ---- A.2 undistortPoints: 5 fixed-point iterations --------------------------------------------- */
void orc_undistort_points(const double* img, int n, const double K[9], int dist_model,
                          const double D[8], double* out)
{
  const double fx = K[0], cx = K[2], fy = K[4], cy = K[5];
  for (int i = 0; i < n; ++i) {
    double x0 = (img[2 * i] - cx) / fx, y0 = (img[2 * i + 1] - cy) / fy;
    double x = x0, y = y0;
    if (dist_model == RCC_DIST_PLUMB_BOB) {
      const double k1 = D[0], k2 = D[1], p1 = D[2], p2 = D[3], k3 = D[4];
      for (int it = 0; it < 5; ++it) {
        double r2 = x * x + y * y;
        double icdist = 1.0 / (1.0 + ((k3 * r2 + k2) * r2 + k1) * r2);
        double dx = 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x);
        double dy = p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y;
        x = (x0 - dx) * icdist;
        y = (y0 - dy) * icdist;
      }
    }
    out[2 * i] = x;
    out[2 * i + 1] = y;
  }
}

/* ---- A.7 projectPoints + Jacobian ---------------------------------------------------------------- */
void orc_project_points(const double* obj, int n, const double r[3], const double t[3],
                        const double K[9], int dist_model, const double D[8], double* uv,
                        double* dpdr, double* dpdt)
{
  double R[9], dRdr[27];
  orc_rodrigues_v2m(r, R, dRdr);
  const double fx = K[0], cx = K[2], fy = K[4], cy = K[5];
  double k[5] = { 0, 0, 0, 0, 0 };
  if (dist_model == RCC_DIST_PLUMB_BOB) for (int i = 0; i < 5; ++i) k[i] = D[i];
  for (int i = 0; i < n; ++i) {
    double X = obj[3 * i], Y = obj[3 * i + 1], Z = obj[3 * i + 2];
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
    uv[2 * i] = xd * fx + cx;
    uv[2 * i + 1] = yd * fy + cy;
    if (dpdt) {
      double dxdt[3] = { z, 0, -x * z }, dydt[3] = { 0, z, -y * z };
      for (int j = 0; j < 3; ++j) {
        double dr2 = 2.0 * x * dxdt[j] + 2.0 * y * dydt[j];
        double dcd = k[0] * dr2 + 2.0 * k[1] * r2 * dr2 + 3.0 * k[4] * r4 * dr2;
        double da1 = 2.0 * (x * dydt[j] + y * dxdt[j]);
        dpdt[(2 * i) * 3 + j] = fx * (dxdt[j] * cdist + x * dcd + k[2] * da1 + k[3] * (dr2 + 4.0 * x * dxdt[j]));
        dpdt[(2 * i + 1) * 3 + j] = fy * (dydt[j] * cdist + y * dcd + k[2] * (dr2 + 4.0 * y * dydt[j]) + k[3] * da1);
      }
    }
    if (dpdr) {
      for (int j = 0; j < 3; ++j) {
        const double* d = dRdr + j * 9;
        double dx0 = X * d[0] + Y * d[1] + Z * d[2];
        double dy0 = X * d[3] + Y * d[4] + Z * d[5];
        double dz0 = X * d[6] + Y * d[7] + Z * d[8];
        double dxdr = z * (dx0 - x * dz0), dydr = z * (dy0 - y * dz0);
        double dr2 = 2.0 * x * dxdr + 2.0 * y * dydr;
        double dcd = k[0] * dr2 + 2.0 * k[1] * r2 * dr2 + 3.0 * k[4] * r4 * dr2;
        double da1 = 2.0 * (x * dydr + y * dxdr);
        dpdr[(2 * i) * 3 + j] = fx * (dxdr * cdist + x * dcd + k[2] * da1 + k[3] * (dr2 + 4.0 * x * dxdr));
        dpdr[(2 * i + 1) * 3 + j] = fy * (dydr * cdist + y * dcd + k[2] * (dr2 + 4.0 * y * dydr) + k[3] * da1);
      }
    }
  }
}

/* ---- A.4 homography: normalised DLT on float32 points (+ LM refinement for N > 4) --------------- */
static void homography_residual(const double h[8], const float* M, const float* m, int n,
                                double* err, double* J /* 2n x 8 or NULL */)
{
  for (int i = 0; i < n; ++i) {
    double Mx = M[2 * i], My = M[2 * i + 1];
    double ww = h[6] * Mx + h[7] * My + 1.0;
    ww = fabs(ww) > DBL_EPSILON ? 1.0 / ww : 0.0;
    double xi = (h[0] * Mx + h[1] * My + h[2]) * ww;
    double yi = (h[3] * Mx + h[4] * My + h[5]) * ww;
    err[2 * i] = xi - (double)m[2 * i];
    err[2 * i + 1] = yi - (double)m[2 * i + 1];
    if (J) {
      double* a = J + (2 * i) * 8;
      double* b = a + 8;
      a[0] = Mx * ww; a[1] = My * ww; a[2] = ww; a[3] = a[4] = a[5] = 0.0;
      a[6] = -Mx * ww * xi; a[7] = -My * ww * xi;
      b[0] = b[1] = b[2] = 0.0; b[3] = Mx * ww; b[4] = My * ww; b[5] = ww;
      b[6] = -Mx * ww * yi; b[7] = -My * ww * yi;
    }
  }
}

/* The LMSolver run used by findHomography(method 0) for N > 4, at most 10 iterations
 * ([U], medium confidence: structure after OpenCV 3.4 calib3d/levmarq.cpp as recalled). */
static void homography_refine(double h[8], const float* M, const float* m, int n)
{
  const int P = 8, maxIters = 10;
  const double epsx = FLT_EPSILON, epsf = FLT_EPSILON;
  const int m2 = 2 * n;
  double* r = (double*)malloc(sizeof(double) * m2);
  double* rd = (double*)malloc(sizeof(double) * m2);
  double* J = (double*)malloc(sizeof(double) * m2 * P);
  double x[8], xd[8], A[64], Ap[64], v[8], d[8], Dg[8], tmp[8];
  memcpy(x, h, sizeof(x));
  homography_residual(x, M, m, n, r, J);
  double S = 0.0;
  for (int i = 0; i < m2; ++i) S += r[i] * r[i];
#define JTJ()                                                                      \
  do {                                                                             \
    for (int a_ = 0; a_ < P; ++a_) {                                               \
      for (int b_ = 0; b_ < P; ++b_) {                                             \
        double s_ = 0.0;                                                           \
        for (int i_ = 0; i_ < m2; ++i_) s_ += J[i_ * P + a_] * J[i_ * P + b_];     \
        A[a_ * P + b_] = s_;                                                       \
      }                                                                            \
      double g_ = 0.0;                                                             \
      for (int i_ = 0; i_ < m2; ++i_) g_ += J[i_ * P + a_] * r[i_];                \
      v[a_] = g_;                                                                  \
    }                                                                              \
  } while (0)
  JTJ();
  for (int i = 0; i < P; ++i) Dg[i] = A[i * P + i];
  const double Rlo = 0.25, Rhi = 0.75;
  double lambda = 1.0, lc = 0.75;
  int iter = 0;
  for (;;) {
    memcpy(Ap, A, sizeof(A));
    for (int i = 0; i < P; ++i) Ap[i * P + i] += lambda * Dg[i];
    sym_solve(P, Ap, v, d);
    for (int i = 0; i < P; ++i) xd[i] = x[i] - d[i];
    homography_residual(xd, M, m, n, rd, NULL);
    double Sd = 0.0;
    for (int i = 0; i < m2; ++i) Sd += rd[i] * rd[i];
    /* temp_d = -A d + 2 v ; dS = d . temp_d */
    double dS = 0.0;
    for (int a = 0; a < P; ++a) {
      double s = 0.0;
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
      for (int a = 0; a < P; ++a) t += d[a] * v[a];
      double nu = (Sd - S) / (fabs(t) > DBL_EPSILON ? t : 1.0) + 2.0;
      nu = nu < 2.0 ? 2.0 : (nu > 10.0 ? 10.0 : nu);
      if (lambda == 0.0) {
        /* 1 / max |diag(inv(A))| via the eigen-decomposition */
        double T[64], w[8], V[64];
        memcpy(T, A, sizeof(A));
        orc_jacobi_eigen_sym(P, T, w, V);
        double thr = 0.0;
        for (int i = 0; i < P; ++i) thr += fabs(w[i]);
        thr *= 2.0 * DBL_EPSILON;
        double maxval = DBL_EPSILON;
        for (int a = 0; a < P; ++a) {
          double s = 0.0;
          for (int i = 0; i < P; ++i) if (fabs(w[i]) > thr) s += V[i * P + a] * V[i * P + a] / w[i];
          if (fabs(s) > maxval) maxval = fabs(s);
        }
        lambda = lc = 1.0 / maxval;
        nu *= 0.5;
      }
      lambda *= nu;
    }
    if (Sd < S) {
      S = Sd;
      memcpy(x, xd, sizeof(x));
      homography_residual(x, M, m, n, r, J);
      JTJ();
    }
    ++iter;
    double dinf = 0.0, rinf = 0.0;
    for (int i = 0; i < P; ++i) if (fabs(d[i]) > dinf) dinf = fabs(d[i]);
    for (int i = 0; i < m2; ++i) if (fabs(r[i]) > rinf) rinf = fabs(r[i]);
    if (!(iter < maxIters && dinf >= epsx && rinf >= epsf)) break;
  }
#undef JTJ
  memcpy(h, x, sizeof(x));
  free(r); free(rd); free(J);
}

int orc_find_homography(const double* src, const double* dst, int n, double H[9])
{
  float* M = (float*)malloc(sizeof(float) * 2 * n);
  float* m = (float*)malloc(sizeof(float) * 2 * n);
  for (int i = 0; i < 2 * n; ++i) { M[i] = (float)src[i]; m[i] = (float)dst[i]; }
  double cMx = 0, cMy = 0, cmx = 0, cmy = 0;
  for (int i = 0; i < n; ++i) { cMx += M[2 * i]; cMy += M[2 * i + 1]; cmx += m[2 * i]; cmy += m[2 * i + 1]; }
  cMx /= n; cMy /= n; cmx /= n; cmy /= n;
  double sMx = 0, sMy = 0, smx = 0, smy = 0;
  for (int i = 0; i < n; ++i) {
    sMx += fabs(M[2 * i] - cMx); sMy += fabs(M[2 * i + 1] - cMy);
    smx += fabs(m[2 * i] - cmx); smy += fabs(m[2 * i + 1] - cmy);
  }
  int ok = 1;
  if (fabs(sMx) < DBL_EPSILON || fabs(sMy) < DBL_EPSILON || fabs(smx) < DBL_EPSILON || fabs(smy) < DBL_EPSILON) ok = 0;
  if (ok) {
    smx = n / smx; smy = n / smy; sMx = n / sMx; sMy = n / sMy;
    double invHnorm[9] = { 1.0 / smx, 0, cmx, 0, 1.0 / smy, cmy, 0, 0, 1 };
    double Hnorm2[9] = { sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1 };
    double LtL[81], w[9], V[81];
    memset(LtL, 0, sizeof(LtL));
    for (int i = 0; i < n; ++i) {
      double x = (m[2 * i] - cmx) * smx, y = (m[2 * i + 1] - cmy) * smy;
      double X = (M[2 * i] - cMx) * sMx, Y = (M[2 * i + 1] - cMy) * sMy;
      double Lx[9] = { X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x };
      double Ly[9] = { 0, 0, 0, X, Y, 1, -y * X, -y * Y, -y };
      for (int j = 0; j < 9; ++j)
        for (int k = j; k < 9; ++k) LtL[j * 9 + k] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
    }
    for (int j = 0; j < 9; ++j) for (int k = 0; k < j; ++k) LtL[j * 9 + k] = LtL[k * 9 + j];
    orc_jacobi_eigen_sym(9, LtL, w, V);
    double H0[9], T[9];
    for (int k = 0; k < 9; ++k) H0[k] = V[8 * 9 + k];
    mat3_mul(invHnorm, H0, T);
    mat3_mul(T, Hnorm2, H);
    double s = 1.0 / H[8];
    for (int k = 0; k < 9; ++k) H[k] *= s;
    H[8] = 1.0;
    if (n > 4) {
      double h8[8];
      memcpy(h8, H, sizeof(h8));
      homography_refine(h8, M, m, n);
      memcpy(H, h8, sizeof(h8));
      H[8] = 1.0;
    }
    for (int k = 0; k < 9; ++k) if (!isfinite(H[k])) ok = 0;
  }
  free(M); free(m);
  return ok;
}

/* ---- A.1, A.3, A.5, A.8: the solver ---------------------------------------------------------------- */
int orc_solve_pnp(const double* obj, const double* img, int n, const double K[9], int dist_model,
                  const double D[8], double rvec[3], double tvec[3], double* rms, int* iters_out)
{
  if (rms) *rms = 0.0;
  if (iters_out) *iters_out = 0;
  rvec[0] = rvec[1] = rvec[2] = 0.0;
  tvec[0] = tvec[1] = tvec[2] = 0.0;
  if (n < 4) return RCC_PNP_TOO_FEW;
  double* mn = (double*)malloc(sizeof(double) * 2 * n);
  double* Mxy = (double*)malloc(sizeof(double) * 2 * n);
  double* J = (double*)malloc(sizeof(double) * 2 * n * 6);
  double* dpdr = (double*)malloc(sizeof(double) * 2 * n * 3);
  double* dpdt = (double*)malloc(sizeof(double) * 2 * n * 3);
  double* err = (double*)malloc(sizeof(double) * 2 * n);
  double* uv = (double*)malloc(sizeof(double) * 2 * n);
  int status = RCC_PNP_OK;

  orc_undistort_points(img, n, K, dist_model, D, mn);

  /* A.3 */
  double Mc[3] = { 0, 0, 0 };
  for (int i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) Mc[k] += obj[3 * i + k];
  for (int k = 0; k < 3; ++k) Mc[k] /= n;
  double MM[9];
  memset(MM, 0, sizeof(MM));
  for (int i = 0; i < n; ++i) {
    double d[3] = { obj[3 * i] - Mc[0], obj[3 * i + 1] - Mc[1], obj[3 * i + 2] - Mc[2] };
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) MM[a * 3 + b] += d[a] * d[b];
  }
  double W[3], Vt[9];
  orc_jacobi_eigen_sym(3, MM, W, Vt);
  double R[9], t[3], r[3];
  if (!(W[2] / W[1] < 1e-3)) {
    status = RCC_PNP_NONPLANAR;
    goto done;
  }
  {
    double Rt[9];
    memcpy(Rt, Vt, sizeof(Rt));
    if (Vt[2] * Vt[2] + Vt[5] * Vt[5] < 1e-10) {
      for (int k = 0; k < 9; ++k) Rt[k] = (k % 4 == 0) ? 1.0 : 0.0;
    }
    if (mat3_det(Rt) < 0) for (int k = 0; k < 9; ++k) Rt[k] = -Rt[k];
    double Tt[3];
    for (int a = 0; a < 3; ++a) Tt[a] = -(Rt[a * 3] * Mc[0] + Rt[a * 3 + 1] * Mc[1] + Rt[a * 3 + 2] * Mc[2]);
    for (int i = 0; i < n; ++i) {
      const double* M = obj + 3 * i;
      Mxy[2 * i] = Rt[0] * M[0] + Rt[1] * M[1] + Rt[2] * M[2] + Tt[0];
      Mxy[2 * i + 1] = Rt[3] * M[0] + Rt[4] * M[1] + Rt[5] * M[2] + Tt[1];
    }
    double H[9];
    if (orc_find_homography(Mxy, mn, n, H)) {
      double h1[3] = { H[0], H[3], H[6] }, h2[3] = { H[1], H[4], H[7] };
      t[0] = H[2]; t[1] = H[5]; t[2] = H[8];
      double n1 = sqrt(h1[0] * h1[0] + h1[1] * h1[1] + h1[2] * h1[2]);
      double n2 = sqrt(h2[0] * h2[0] + h2[1] * h2[1] + h2[2] * h2[2]);
      for (int k = 0; k < 3; ++k) { h1[k] /= n1; h2[k] /= n2; t[k] *= 2.0 / (n1 + n2); }
      double h3[3] = { h1[1] * h2[2] - h1[2] * h2[1], h1[2] * h2[0] - h1[0] * h2[2], h1[0] * h2[1] - h1[1] * h2[0] };
      double R0[9] = { h1[0], h2[0], h3[0], h1[1], h2[1], h3[1], h1[2], h2[2], h3[2] };
      orc_rodrigues_m2v(R0, r);
      orc_rodrigues_v2m(r, R, NULL);
      double t2[3];
      for (int a = 0; a < 3; ++a) t2[a] = R[a * 3] * Tt[0] + R[a * 3 + 1] * Tt[1] + R[a * 3 + 2] * Tt[2] + t[a];
      memcpy(t, t2, sizeof(t));
      mat3_mul(R, Rt, R);
    } else {
      status = RCC_PNP_DEGENERATE;
      for (int k = 0; k < 9; ++k) R[k] = (k % 4 == 0) ? 1.0 : 0.0;
      t[0] = t[1] = t[2] = 0.0;
    }
    orc_rodrigues_m2v(R, r);
  }

  /* A.8 CvLevMarq, the (J, err) update form, 6 parameters, all free */
  {
    double p[6] = { r[0], r[1], r[2], t[0], t[1], t[2] }, pprev[6];
    double A[36], g[6], Ap[36], dl[6];
    int L = -3, it = 0;
    double prevErr = 0.0;
    const int max_iter = 20;
    const double eps = FLT_EPSILON;
    for (;;) {
      /* state CALC_J: J and err at p */
      orc_project_points(obj, n, p, p + 3, K, dist_model, D, uv, dpdr, dpdt);
      for (int i = 0; i < 2 * n; ++i) {
        err[i] = uv[i] - img[i];
        for (int k = 0; k < 3; ++k) { J[i * 6 + k] = dpdr[i * 3 + k]; J[i * 6 + 3 + k] = dpdt[i * 3 + k]; }
      }
      for (int a = 0; a < 6; ++a) {
        for (int b = 0; b < 6; ++b) {
          double s = 0.0;
          for (int i = 0; i < 2 * n; ++i) s += J[i * 6 + a] * J[i * 6 + b];
          A[a * 6 + b] = s;
        }
        double s = 0.0;
        for (int i = 0; i < 2 * n; ++i) s += J[i * 6 + a] * err[i];
        g[a] = s;
      }
      memcpy(pprev, p, sizeof(p));
      if (it == 0) {
        double s = 0.0;
        for (int i = 0; i < 2 * n; ++i) s += err[i] * err[i];
        prevErr = sqrt(s);
      }
      double errNorm;
      for (;;) {
        /* step(): p = pprev - solve(A with diag*(1+lambda), g) */
        double lambda = exp((double)L * log(10.0));
        memcpy(Ap, A, sizeof(A));
        for (int a = 0; a < 6; ++a) Ap[a * 6 + a] *= 1.0 + lambda;
        sym_solve(6, Ap, g, dl);
        for (int a = 0; a < 6; ++a) p[a] = pprev[a] - dl[a];
        /* state CHECK_ERR */
        orc_project_points(obj, n, p, p + 3, K, dist_model, D, uv, NULL, NULL);
        double s = 0.0;
        for (int i = 0; i < 2 * n; ++i) { double e = uv[i] - img[i]; s += e * e; }
        errNorm = sqrt(s);
        if (errNorm > prevErr) {
          if (++L <= 16) continue;
        }
        break;
      }
      L = (L - 1 > -16) ? L - 1 : -16;
      double dn = 0.0, pn = 0.0;
      for (int a = 0; a < 6; ++a) { dn += (p[a] - pprev[a]) * (p[a] - pprev[a]); pn += pprev[a] * pprev[a]; }
      /* cvNorm(param, prevParam, CV_RELATIVE_L2) = |param - prev| / |prev| */
      double rel = sqrt(dn) / sqrt(pn);
      if (++it >= max_iter || rel < eps) break;
      prevErr = errNorm;
    }
    memcpy(rvec, p, sizeof(double) * 3);
    memcpy(tvec, p + 3, sizeof(double) * 3);
    if (iters_out) *iters_out = it;
    if (rms) {
      orc_project_points(obj, n, rvec, tvec, K, dist_model, D, uv, NULL, NULL);
      double s = 0.0;
      for (int i = 0; i < 2 * n; ++i) { double e = uv[i] - img[i]; s += e * e; }
      *rms = sqrt(s / n);
    }
  }
done:
  free(mn); free(Mxy); free(J); free(dpdr); free(dpdt); free(err); free(uv);
  return status;
}
