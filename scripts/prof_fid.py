#!/usr/bin/env python3
"""PMC workload for the fiducial path: B 1080p frames of a 6x4 tag grid through the whole path, 2 times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robot_camera_calibration_amd import abi, api, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = api.default_config(); abi.set_geometry(cfg, 1920, 1080)
family = abi.load_family(); abi.set_fiducial_target(cfg, family, tag_size=0.10, max_targets=24)
cfg.batch_capacity = B
if len(sys.argv) > 2: cfg.max_kept = int(sys.argv[2])
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
for _ in range(2):
    d, fc = det.detect(frames, B, want_corners=False)
print("detections", len(d))
