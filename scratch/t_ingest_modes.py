#!/usr/bin/env python3
"""the ingest pass under different cameras: default plumb-bob, no distortion (identity map through the staged kernel),
and undistort off (the streaming grey conversion k_grey_bgr_stream: the bound of any BGR -> grey pass)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robot_camera_calibration_amd import abi, api
B = 1024
for name, mod in (("plumb-bob default", lambda c: None),
                  ("identity map (D = 0)", lambda c: abi.set_distortion(c, abi.RCC_DIST_PLUMB_BOB, (0.0, 0.0, 0.0, 0.0, 0.0))),
                  ("model none", lambda c: abi.set_distortion(c, abi.RCC_DIST_NONE, ())),
                  ("undistort off (streaming grey)", lambda c: setattr(c, "undistort", 0))):
    cfg = api.default_config(); abi.set_geometry(cfg, 1920, 1080); cfg.batch_capacity = B
    mod(cfg)
    det = api.Detector(cfg)
    frames = torch.randint(0, 255, (B, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
    grey = torch.empty((B, 1920 * 1080), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    det.time_ingest(frames, B, grey, 2)
    r = [det.time_ingest(frames, B, grey, 6) for _ in range(3)]
    print("%-34s %s ms -> %.0f GB/s on 4 px" % (name, ["%.3f" % x for x in r], 4 * 1920 * 1080 * B / (min(r) * 1e-3) / 1e9), flush=True)
    det.close(); del frames, grey; torch.cuda.empty_cache()
