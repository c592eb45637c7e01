#!/usr/bin/env python3
"""Why is the compact threshold+corner pass 1.41 ms per 256 4K frames but 1.08 ms per 1024 1080p frames (same pixel count)?
Times both geometries with the flat-row skip on and off."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robot_camera_calibration_amd import abi, api, synth
for (W, H, B) in [(1920, 1080, 1024), (3840, 2160, 256)]:
    cfg = api.default_config(); abi.set_geometry(cfg, W, H); cfg.batch_capacity = B
    det = api.Detector(cfg)
    frames = torch.empty((B, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
    sp = abi.default_synth_params(); poses = synth.sample_poses(32, cfg)
    poses = np.concatenate([poses] * ((B + 31) // 32))[:B]
    for s0 in range(0, B, 32):
        det.synth_render(sp, poses[s0:s0 + 32], frames[s0:s0 + 32], first_index=s0)
    px = W * H
    grey = torch.empty((B, px), dtype=torch.uint8, device="cuda:0"); binm = torch.empty_like(grey)
    cand = torch.empty((B, cfg.max_candidates * 8), dtype=torch.uint8, device="cuda:0"); cnt = torch.empty((B,), dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    det.stage_ingest(frames, B, grey); torch.cuda.synchronize()
    out = []
    for skip in (1, 0):
        det.set_dense_skip(skip)
        for form, b in (("stage", binm), ("compact", None)):
            det.time_dense(grey, B, b, cand, cnt, 2)
            out.append("%s skip=%d %.3f ms" % (form, skip, det.time_dense(grey, B, b, cand, cnt, 5)))
    det.stage_threshold_corner(grey, B, binm, cand, cnt); torch.cuda.synchronize()
    print("%dx%d x %d: " % (W, H, B) + "; ".join(out) + "; candidates per frame %.0f" % cnt.float().mean().item())
    det.close(); del frames, grey, binm, cand, cnt; torch.cuda.empty_cache()
