#!/usr/bin/env python3
"""per-stage times of the synchronous step (events inside rcc_detect_batch) against batch size: us per frame"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robot_camera_calibration_amd import abi, api, synth
for B in (1024, 1536, 2048, 2560, 3072, 4096):
    cfg = api.default_config(); abi.set_geometry(cfg, 1920, 1080); cfg.batch_capacity = B
    det = api.Detector(cfg)
    frames = torch.empty((B, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
    poses = synth.sample_poses(B, cfg)
    for s0 in range(0, B, 64):
        det.synth_render(abi.default_synth_params(), poses[s0:s0 + 64], frames[s0:s0 + 64], first_index=s0)
    torch.cuda.synchronize()
    acc = np.zeros(5)
    for r in range(6):
        det.detect(frames, B, want_corners=False)
        if r: acc += np.array(list(det.last_timings().values()))
    acc /= 5
    print("batch %5d: ingest %.3f dense %.3f list+subpix+grid %.3f pnp %.3f d2h %.3f us per frame" % ((B,) + tuple(1e3 * acc / B)), flush=True)
    det.close(); del frames; torch.cuda.empty_cache()
