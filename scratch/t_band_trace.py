#!/usr/bin/env python3
"""run the compact threshold+corner pass once on B bench frames (library built with -DRCC_BAND_TRACE prints per-wave timers)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robot_camera_calibration_amd import abi, api, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = api.default_config(); abi.set_geometry(cfg, 1920, 1080); cfg.batch_capacity = B
det = api.Detector(cfg)
frames = torch.empty((B, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
sp = abi.default_synth_params(); poses = synth.sample_poses(B, cfg)
for s0 in range(0, B, 64):
    det.synth_render(sp, poses[s0:s0 + 64], frames[s0:s0 + 64], first_index=s0)
px = 1920 * 1080
grey = torch.empty((B, px), dtype=torch.uint8, device="cuda:0")
cand = torch.empty((B, cfg.max_candidates * 8), dtype=torch.uint8, device="cuda:0"); cnt = torch.empty((B,), dtype=torch.int32, device="cuda:0")
torch.cuda.synchronize(); det.stage_ingest(frames, B, grey); torch.cuda.synchronize()
print("B", B, "compact %.3f ms" % det.time_dense(grey, B, None, cand, cnt, 1))
