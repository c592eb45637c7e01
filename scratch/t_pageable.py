#!/usr/bin/env python3
"""Host-resident batch: pinned against truly pageable input through the chunked copy / compute pipeline and the one-copy form
(the pipeline queued all copies before any kernel; pageable copies hold the host, so nothing overlapped).
1024 x 1080p BGR8, median of 5."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from robot_camera_calibration_amd import abi, api, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = api.default_config(); abi.set_geometry(cfg, 1920, 1080, abi.RCC_PIX_BGR8); cfg.batch_capacity = B
det = api.Detector(cfg)
sp = abi.default_synth_params(); poses = synth.sample_poses(B, cfg)
frames = torch.empty((B, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
for s0 in range(0, B, 64): det.synth_render(sp, poses[s0:s0 + 64], frames[s0:s0 + 64], first_index=s0)
pinned = torch.empty((B, cfg.frame_bytes), dtype=torch.uint8, pin_memory=True); pinned.copy_(frames); torch.cuda.synchronize()
pageable = np.array(pinned.numpy(), copy=True)
def timed(fn, reps=5):
    fn(); ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return statistics.median(ts)
for name, buf in (("pinned", pinned), ("pageable", pageable)):
    for chunk, cname in ((0, "pipeline"), (-1, "one copy")):
        det.set_host_chunk(chunk)
        t = timed(lambda: det.detect(buf, B, want_corners=False))
        print("%-9s %-9s %8.1f ms  %7.0f frames/s  %5.1f GB/s" % (name, cname, 1e3 * t, B / t, B * cfg.frame_bytes / t / 1e9), flush=True)
