#!/usr/bin/env python3
"""Workload for the PMC passes on the threshold+corner kernel: B 1080p frames, 3 launches of the
dense pass and 3 calibration copies of the same grey buffer (known byte count, same access width).
Run under:  rocprofv3 --kernel-trace --pmc FETCH_SIZE  (and again with WRITE_SIZE)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robot_camera_calibration_amd import abi, api, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = api.default_config(); abi.set_geometry(cfg, 1920, 1080); cfg.batch_capacity = B
det = api.Detector(cfg)
if len(sys.argv) > 3: det.set_dense_gang(int(sys.argv[2]), int(sys.argv[3]))
frames = torch.empty((B, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
sp = abi.default_synth_params(); poses = synth.sample_poses(32, cfg)
poses = np.concatenate([poses] * ((B + 31) // 32))[:B]
for s0 in range(0, B, 64):
    n = min(64, B - s0)
    det.synth_render(sp, poses[s0:s0 + n], frames[s0:s0 + n], first_index=s0)
px = 1920 * 1080
grey = torch.empty((B, px), dtype=torch.uint8, device="cuda:0"); binm = torch.empty_like(grey)
cand = torch.empty((B, cfg.max_candidates * 8), dtype=torch.uint8, device="cuda:0"); cnt = torch.empty((B,), dtype=torch.int32, device="cuda:0")
torch.cuda.synchronize()
det.stage_ingest(frames, B, grey)
for _ in range(3):
    det.stage_threshold_corner(grey, B, binm, cand, cnt)
det.time_dense(grey, B, None, cand, cnt, 3)      # the form rcc_detect_batch launches: compact threshold map
for _ in range(3):
    det._chk(det._L.rcc_debug_calib_copy(det._h, api._ptr(grey), api._ptr(binm), B * px), "calib")
print("frames", B, "alg bytes per dense launch", 2 * px * B, "copy bytes read", B * px, "written", B * px)
