#!/usr/bin/env python3
"""the ingest pass alone: ms per B 1080p frames (back-to-back launches)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robot_camera_calibration_amd import abi, api, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = api.default_config(); abi.set_geometry(cfg, 1920, 1080); cfg.batch_capacity = B
det = api.Detector(cfg)
if len(sys.argv) > 2: det.set_ingest_variant(int(sys.argv[2]))
frames = torch.randint(0, 255, (B, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
grey = torch.empty((B, 1920 * 1080), dtype=torch.uint8, device="cuda:0")
torch.cuda.synchronize()
det.time_ingest(frames, B, grey, 2)
r = [det.time_ingest(frames, B, grey, 6) for _ in range(3)]
print("ingest %d x 1080p: %s ms -> %.0f GB/s on 4 px" % (B, ["%.3f" % x for x in r], 4 * 1920 * 1080 * B / (min(r) * 1e-3) / 1e9))
