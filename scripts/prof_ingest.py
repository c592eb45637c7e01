#!/usr/bin/env python3
"""PMC workload: B 1080p frames, 3 launches of the ingest pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robot_camera_calibration_amd import abi, api, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = api.default_config(); abi.set_geometry(cfg, 1920, 1080); cfg.batch_capacity = B
det = api.Detector(cfg)
if len(sys.argv) > 2: det.set_ingest_variant(int(sys.argv[2]))
frames = torch.randint(0, 255, (B, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
grey = torch.empty((B, 1920 * 1080), dtype=torch.uint8, device="cuda:0")
torch.cuda.synchronize()
for _ in range(3):
    det.stage_ingest(frames, B, grey)
g2 = torch.empty_like(grey)
torch.cuda.synchronize()
for _ in range(3):      # calibration copies: B * 1920*1080 bytes each way
    det._chk(det._L.rcc_debug_calib_copy(det._h, api._ptr(grey), api._ptr(g2), B * 1920 * 1080), "calib")
print("done")
