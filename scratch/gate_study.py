#!/usr/bin/env python3
"""a5's gate under heavy blur: for the candidates nearest the ground-truth corners, the transitions their ring shows at several radii,
and how far the Harris maximum sits from the corner."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from robot_camera_calibration_amd import abi, api, synth
n = 256
cfg = api.default_config(); abi.set_geometry(cfg, 1920, 1080, abi.RCC_PIX_BGR8); cfg.batch_capacity = n
cfg.xj_check = 0                      # refine and keep everything: the study looks at the candidates themselves
det = api.Detector(cfg)
poses = synth.sample_poses(n, cfg); objb = synth.board_object_points(8, 6, 0.108); K = np.array(list(cfg.K))
frames = torch.empty((n, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
def ring(R):
    a = 2 * np.pi * np.arange(16) / 16
    return np.stack([np.rint(R * np.cos(a)), np.rint(R * np.sin(a))], 1).astype(int)
RADII = (8, 9, 10, 11, 12)
RG = {R: ring(R) for R in RADII}
for blur in (1.5, 2.0, 2.3):
    sp = abi.set_optics(abi.default_synth_params(), blur, 0, 0, 0)
    for s0 in range(0, n, 64): det.synth_render(sp, poses[s0:s0 + 64], frames[s0:s0 + 64], first_index=s0)
    det.detect(frames, n)
    img = det.fetch_images(n); lst = det.fetch_lists(n)
    offs = []; fails = {R: 0 for R in RADII}; tot = 0; nocand = 0; fail_frames = {R: set() for R in RADII}
    for f in range(n):
        g = img["grey"][f]; pre = lst["pre"][f][:lst["npre"][f]]
        gt = synth.project_points(objb, poses[f][:3], poses[f][3:], K)
        for c in gt:
            d = np.hypot(pre["x"] - c[0], pre["y"] - c[1])
            if len(d) == 0 or d.min() > 7: nocand += 1; continue
            # the strongest candidate within 7 px is the one that survives de-duplication
            m = np.flatnonzero(d <= 7); i = m[np.argmax(pre["score"][m])]
            x, y = int(pre["x"][i]), int(pre["y"][i]); offs.append(d[i]); tot += 1
            for R in RADII:
                rg = RG[R]
                if x < R or y < R or x >= 1920 - R or y >= 1080 - R: continue
                v = g[y + rg[:, 1], x + rg[:, 0]].astype(int)
                b = v > ((v.min() + v.max()) >> 1)
                if v.max() - v.min() < 16 or (b != np.roll(b, -1)).sum() < 4: fails[R] += 1; fail_frames[R].add(f)
    offs = np.array(offs)
    print("blur %.1f: corners with a candidate %d (none within 7 px: %d); offset of that candidate: median %.2f p99 %.2f max %.2f; held back by the gate at radius %s; frames touched %s" % (
        blur, tot, nocand, np.median(offs), np.percentile(offs, 99), offs.max(), {R: fails[R] for R in RADII}, {R: len(fail_frames[R]) for R in RADII}), flush=True)
