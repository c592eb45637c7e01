#!/usr/bin/env python3
"""Threshold+corner pass, fused band kernel (variant 1) against band sweep + corner kernel on the active rows (variant 3):
mean ms per 1024 x 1080p launch, stage form (full binary image) and compact form (per-tile threshold map)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robot_camera_calibration_amd import abi, api, synth
W, H = 1920, 1080
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = api.default_config(); abi.set_geometry(cfg, W, H); cfg.batch_capacity = B
det = api.Detector(cfg)
px = W * H
grey = torch.empty((B, px), dtype=torch.uint8, device="cuda:0"); binm = torch.empty_like(grey)
cand = torch.empty((B, cfg.max_candidates * 8), dtype=torch.uint8, device="cuda:0"); cnt = torch.empty((B,), dtype=torch.int32, device="cuda:0")
frames = torch.empty((B, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
sp = abi.default_synth_params(); poses = synth.sample_poses(32, cfg)
poses = np.concatenate([poses] * ((B + 31) // 32))[:B]
for s0 in range(0, B, 64):
    det.synth_render(sp, poses[s0:s0 + 64], frames[s0:s0 + 64], first_index=s0)
torch.cuda.synchronize()
det.stage_ingest(frames, B, grey)
ref = None
for variant in [int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else "1,3,1,3".split(","))]:
    det.set_dense_variant(variant)
    out = []
    for form, b in (("stage", binm), ("compact", None)):
        det.time_dense(grey, B, b, cand, cnt, 2)
        t = det.time_dense(grey, B, b, cand, cnt, reps)
        gb = (2 * px if b is not None else px * 17 / 16) * B / 1e9
        out.append("%s %.3f ms (%.0f GB/s own bytes, %.3f of 8 TB/s on 2 px)" % (form, t, gb / t * 1e3, 2 * px * B / 1e9 / t * 1e3 / 8000))
    k = cnt.cpu().numpy().copy()
    if ref is None: ref = k
    print("variant %d: %s; candidates %d %s" % (variant, "; ".join(out), int(k.sum()), "same counts" if (k == ref).all() else "COUNTS DIFFER"), flush=True)
