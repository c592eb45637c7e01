#!/usr/bin/env python3
"""PMC workload for the board PnP kernel: B 1080p frames through the whole path, 3 times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robot_camera_calibration_amd import abi, api, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = api.default_config(); abi.set_geometry(cfg, 1920, 1080); cfg.batch_capacity = B
det = api.Detector(cfg)
frames = torch.empty((B, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
sp = abi.default_synth_params(); poses = synth.sample_poses(32, cfg)
poses = np.concatenate([poses] * ((B + 31) // 32))[:B]
for s0 in range(0, B, 64):
    n = min(64, B - s0)
    det.synth_render(sp, poses[s0:s0 + n], frames[s0:s0 + n], first_index=s0)
torch.cuda.synchronize()
for _ in range(3):
    d, fc = det.detect(frames, B)
print("detections", len(d))
