#!/usr/bin/env python3
"""experiment: how much of k_subpix is the dispatch of its empty blocks?  grid.x = max_kept: time the tail with max_kept 256 / 128 / 96"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robot_camera_calibration_amd import abi, api, synth
B = 1024
for mk in (256, 128, 96):
    cfg = api.default_config(); abi.set_geometry(cfg, 1920, 1080); cfg.batch_capacity = B; cfg.max_kept = mk
    det = api.Detector(cfg)
    frames = torch.empty((B, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
    sp = abi.default_synth_params(); poses = synth.sample_poses(64, cfg)
    poses = np.concatenate([poses] * (B // 64))
    for s0 in range(0, B, 64):
        det.synth_render(sp, poses[s0:s0 + 64], frames[s0:s0 + 64], first_index=s0 % 64)
    torch.cuda.synchronize()
    for _ in range(3): d, _ = det.detect(frames, B, want_corners=False)
    ts = []
    for _ in range(5):
        d, _ = det.detect(frames, B, want_corners=False); ts.append(det.last_timings())
    print("max_kept", mk, "found", len(d), {k: round(float(np.median([t[k] for t in ts])), 4) for k in ts[0]})
    det.close()
