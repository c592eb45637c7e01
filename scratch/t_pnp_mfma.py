#!/usr/bin/env python3
"""4-point tag poses (BASELINE.json configs[4]-style batch: 1024 x 1080p frames, 6x4 tags each = 24 576 solves): the vector
form (one lane per target, v_fma_f64) against the matrix-core form (cfg.pnp_use_mfma: JtJ / Jte by v_mfma_f64_16x16x4_f64).
Prints the pose kernel's time per batch (HIP events on the launch stream, rcc_last_timings) and solves/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robot_camera_calibration_amd import abi, api, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = api.default_config(); abi.set_geometry(cfg, 1920, 1080)
family = abi.load_family(); abi.set_fiducial_target(cfg, family, tag_size=0.10, max_targets=24)
cfg.batch_capacity = B
det = api.Detector(cfg)
frames = torch.empty((B, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
sp = abi.default_synth_params()
(fhx, fhy), _, _ = synth.fiducial_grid_layout(6, 4, cfg.tag_size)
sp.fid_grid_x, sp.fid_grid_y, sp.fid_gap_permille = 6, 4, 500
poses = synth.sample_poses(B, cfg, z_range=(1.0, 2.0), max_tilt_deg=40, half_extent_m=(fhx, fhy))
for s0 in range(0, B, 64):
    n = min(64, B - s0)
    det.synth_render(sp, poses[s0:s0 + n], frames[s0:s0 + n], first_index=s0)
torch.cuda.synchronize()
res = {}
for rep in range(2):
    for mf in (0, 1):
        det.set_pnp_mfma(mf)
        det.detect(frames, B, want_corners=False)
        ts = []
        for _ in range(5):
            d, _ = det.detect(frames, B, want_corners=False)
            ts.append(det.last_timings()["pnp"])
        res[mf] = (d, float(np.median(ts)))
        print("pnp_use_mfma %d: pose kernel %.3f ms per batch (median of 5), %d tag poses -> %.2f M solves/s" % (mf, res[mf][1], len(d), len(d) / res[mf][1] / 1e3), flush=True)
a, b = res[0][0], res[1][0]
print("max |rvec| diff %.2e  max |tvec| diff %.2e  iteration counts equal in %.1f %%" % (np.abs(a.rvec - b.rvec).max(), np.abs(a.tvec - b.tvec).max(), 100.0 * (a.pnp_iters == b.pnp_iters).mean()))
