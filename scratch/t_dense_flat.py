#!/usr/bin/env python3
"""Threshold+corner pass on all-flat (constant) frames, on the bench frames, and on all-active (noise) frames: how fast is
the flat path alone (is it the memory system or instruction issue that bounds it)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robot_camera_calibration_amd import abi, api, synth
W, H, B = 1920, 1080, 1024
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
for name in ("flat", "bench", "half", "bottom", "oddframes", "middle", "leftflat", "firsthalf", "every4th"):
    if name == "flat": grey.fill_(128)
    elif name == "bench": det.stage_ingest(frames, B, grey)
    else:
        det.stage_ingest(frames, B, grey); torch.cuda.synchronize()
        g3 = grey.view(B, H, W)
        if name == "half": g3[:, : H // 2, :] = 128       # top half flat
        elif name == "bottom": g3[:, H // 2:, :] = 128
        elif name == "oddframes": g3[1::2] = 128
        elif name == "middle": g3[:, H // 4: 3 * H // 4, :] = 128
        elif name == "leftflat": g3[:, :, : W // 2] = 128
        elif name == "firsthalf": g3[: B // 2] = 128      # the target shows up half way through the batch
        elif name == "every4th":
            keep = torch.arange(B, device="cuda:0") % 4 == 0
            g3[~keep] = 128
    torch.cuda.synchronize()
    out = []
    for form, b in (("stage", binm), ("compact", None)):
        det.time_dense(grey, B, b, cand, cnt, 2)
        t = det.time_dense(grey, B, b, cand, cnt, 5)
        gb = (2 * px if b is not None else px * 17 / 16) * B / 1e9
        out.append("%s %.3f ms (%.0f GB/s)" % (form, t, gb / t * 1e3))
    print(name + ": " + "; ".join(out))
