#!/usr/bin/env python3
"""experiment (librcc_hip_exp.so): where does k_grid_pnp's time go, frame by frame?  The kernel ends with its slowest frame."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RCC_LIBRARY"] = os.path.join(ROOT, "robot_camera_calibration_amd", os.environ.get("RCC_EXP_LIB", "librcc_hip_exp.so"))
import ctypes as C
import numpy as np, torch
from robot_camera_calibration_amd import abi, api, synth
B = 1024
cfg = api.default_config(); abi.set_geometry(cfg, 1920, 1080); cfg.batch_capacity = B
det = api.Detector(cfg)
frames = torch.empty((B, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
sp = abi.default_synth_params(); poses = synth.sample_poses(B, cfg)
for s0 in range(0, B, 64):
    det.synth_render(sp, poses[s0:s0 + 64], frames[s0:s0 + 64], first_index=s0)
torch.cuda.synchronize()
for _ in range(3): det.detect(frames, B, want_corners=False)
tr = torch.zeros((B, 24), dtype=torch.int64, device="cuda:0")
torch.cuda.synchronize()
det._L.rcc_debug_grid_trace.argtypes = [C.c_void_p, C.c_void_p]
assert det._L.rcc_debug_grid_trace(det._h, C.c_void_p(tr.data_ptr())) == 0
d, fc = det.detect(frames, B, want_corners=True)
print("stage ms", det.last_timings())
t = tr.cpu().numpy().astype(np.float64)
t0 = t[:, 0].min()
iters = np.zeros(B, int); iters[d.frame] = d.pnp_iters
us = lambda a: a / 100.0
start = us(t[:, 0] - t0)
ok = t[:, 6] > 0
ph = {"load+seeds": us(t[:, 1] - t[:, 0]), "clear+axes": us(t[:, 3] - t[:, 1]), "growth": us(t[:, 2] - t[:, 3]), "unshear+order": us(t[:, 5] - t[:, 2]), "pose": us(t[:, 6] - t[:, 5]), "pose:plane+norm": us(t[:, 8] - t[:, 5]), "pose:DLT": us(t[:, 9] - t[:, 8]), "pose:H refine": us(t[:, 10] - t[:, 9]), "pose:R,t init": us(t[:, 12] - t[:, 10]), "pose:LM": us(t[:, 6] - t[:, 12]), "total": us(t[:, 6] - t[:, 0])}
print("frames with pose", ok.sum(), " start spread us: min %.1f median %.1f max %.1f" % (start.min(), np.median(start), start.max()))
for k, v in ph.items():
    v = v[ok]
    print("%-14s min %7.1f  median %7.1f  p90 %7.1f  p99 %7.1f  max %7.1f us" % (k, v.min(), np.median(v), np.percentile(v, 90), np.percentile(v, 99), v.max()))
end = us(t[:, 6] - t0)[ok]
print("kernel span (first start -> last end): %.1f us;  median end %.1f" % (end.max(), np.median(end)))
seedatt = (t[:, 7] // 1000).astype(int)
print("seed attempts histogram:", np.bincount(seedatt[ok]))
print("pnp_iters histogram:", np.bincount(iters[ok]))
for i, nm in enumerate(("H acc full", "H acc trial", "H solve8+copy", "H rest", "LM acc full", "LM acc trial", "LM solve6+copy")):
    v = us(t[:, 16 + i])[ok]
    print("   inner %-16s median %6.1f  max %6.1f us" % (nm, np.median(v), v.max()))
print("H refinement iterations histogram:", np.bincount(t[:, 13].astype(int)[ok]))
worst = np.argsort(-ph["total"] * ok)[:8]
for f in worst:
    print("frame %4d total %.1f growth %.1f pose %.1f iters %d seed_attempt %d nkept %d start %.1f" % (f, ph["total"][f], ph["growth"][f], ph["pose"][f], iters[f], seedatt[f], fc[f].nkept, start[f]))
