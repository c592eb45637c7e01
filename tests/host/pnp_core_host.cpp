// Host build of the product's PnP arithmetic (robot_camera_calibration_amd/csrc/pnp_core.h) so its
// logic can be checked on a CPU-only box.  Not the oracle; not used by the product path.
#include "../../robot_camera_calibration_amd/csrc/pnp_core.h"
extern "C" int pnpcore_solve(const double* obj, const double* img, int n, const double* K, int model,
                             const double* D, double* rvec, double* tvec, double* rms, int* iters)
{
  rccpnp::Pts p{ obj, img, n };
  rccpnp::Cam cm;
  cm.fx = K[0]; cm.cx = K[2]; cm.fy = K[4]; cm.cy = K[5];
  for (int i = 0; i < 5; ++i) cm.k[i] = D[i];
  cm.solver = 1;
  double wsl[rccpnp::PNP_WS];
  return rccpnp::solve_pnp(rccpnp::SerialPar{ wsl }, p, cm, model, rvec, tvec, rms, iters);
}
extern "C" void pnpcore_rodrigues_v2m(const double* r, double* R, double* J) { rccpnp::rodrigues_v2m(r, R, J); }
extern "C" void pnpcore_rodrigues_m2v(const double* R, double* r) { rccpnp::rodrigues_m2v(R, r); }

extern "C" void pnpcore_probe(const double* obj, const double* img, int n, const double* K, int model, const double* D, double* out)
{
  rccpnp::Pts p{ obj, img, n };
  rccpnp::Cam cm;
  cm.fx = K[0]; cm.cx = K[2]; cm.fy = K[4]; cm.cy = K[5];
  for (int i = 0; i < 5; ++i) cm.k[i] = D[i];
  cm.solver = 1;
  const bool has_dist = model == RCC_DIST_PLUMB_BOB;
  if (!has_dist) for (int i = 0; i < 5; ++i) cm.k[i] = 0.0;
  double wsl[rccpnp::PNP_WS];
  rccpnp::SerialPar par{ wsl };
  double Rt[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }, Tt[3] = { 0, 0, 0 };
  double H[9];
  int ok = rccpnp::find_homography(par, p, Rt, Tt, cm, has_dist, H);
  for (int i = 0; i < 9; ++i) out[i] = H[i];
  double prm[6];
  int st = rccpnp::pose_init(par, p, cm, has_dist, prm);
  for (int i = 0; i < 6; ++i) out[9 + i] = prm[i];
  double A[36], g[6];
  double S = rccpnp::pose_accumulate(par, prm, p, cm, A, g);
  for (int i = 0; i < 36; ++i) out[15 + i] = A[i];
  for (int i = 0; i < 6; ++i) out[51 + i] = g[i];
  out[57] = S;
  out[58] = (double)(st * 10 + ok);
}
