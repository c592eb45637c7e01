#!/usr/bin/env python3
"""one frame per call (what the ROS node does): wall time per detect() for device- and host-resident input, and -- under
rocprofv3 --kernel-trace -- where the time between the first kernel's start and the last one's end goes"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robot_camera_calibration_amd import abi, api, synth
cfg = api.default_config(); abi.set_geometry(cfg, 1920, 1080); cfg.batch_capacity = 1
det = api.Detector(cfg)
frames = torch.empty((1, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
det.synth_render(abi.default_synth_params(), synth.sample_poses(1, cfg), frames)
torch.cuda.synchronize()
host = torch.empty((1, cfg.frame_bytes), dtype=torch.uint8, pin_memory=True); host.copy_(frames); torch.cuda.synchronize()
for name, src in (("device", frames), ("pinned host", host)):
    for _ in range(20): det.detect(src, 1)
    t = []
    for _ in range(200):
        t0 = time.perf_counter(); d, f = det.detect(src, 1); t.append(time.perf_counter() - t0)
    t = np.array(t) * 1e3
    print("%-12s detect() of one 1080p frame: median %.3f ms, min %.3f, p90 %.3f (found %d)" % (name, np.median(t), t.min(), np.percentile(t, 90), len(d)))
