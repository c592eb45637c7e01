#!/usr/bin/env python3
"""Per-step wall time of the streaming form right after creation: is the first timed step of bench.py slower than the rest?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robot_camera_calibration_amd import abi, api, synth
B = 1024
cfg = api.default_config(); abi.set_geometry(cfg, 1920, 1080); cfg.batch_capacity = B
det = api.Detector(cfg)
frames = torch.empty((B, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
sp = abi.default_synth_params(); poses = synth.sample_poses(32, cfg)
poses = np.concatenate([poses] * 32)[:B]
for s0 in range(0, B, 64):
    det.synth_render(sp, poses[s0:s0 + 64], frames[s0:s0 + 64], first_index=s0)
torch.cuda.synchronize()
det.detect(frames, B, want_corners=False)          # W = 1
torch.cuda.synchronize()
ts = []
t0 = time.perf_counter()
det.submit(frames, B)
for k in range(16):
    if k + 1 < 16: det.submit(frames, B)
    det.collect()
    t1 = time.perf_counter(); ts.append(1e3 * (t1 - t0)); t0 = t1
print("streaming steps (ms):", " ".join("%.2f" % t for t in ts))
ts = []
for k in range(8):
    t0 = time.perf_counter(); det.detect(frames, B, want_corners=False); ts.append(1e3 * (time.perf_counter() - t0))
print("sync steps (ms):", " ".join("%.2f" % t for t in ts))
