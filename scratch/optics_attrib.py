#!/usr/bin/env python3
"""Which stage loses the board under blur / shading: per setting, over the frames, the fraction of the 48 ground-truth corners
that still have (a) a Harris candidate, (b) an entry after list suppression, (c) a validated entry within 3 px."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from robot_camera_calibration_amd import abi, api, synth

n = 64
cfg = api.default_config()
abi.set_geometry(cfg, 1920, 1080, abi.RCC_PIX_BGR8)
cfg.batch_capacity = n
for k, v in [a.split("=") for a in sys.argv[1:]]:
    setattr(cfg, k, int(v))
poses = synth.sample_poses(n, cfg)
objb = synth.board_object_points(8, 6, 0.108)
K = np.array(list(cfg.K))
det = api.Detector(cfg)
frames = torch.empty((n, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
sp0 = abi.default_synth_params()
for blur, sh in [(None, (0, 0, 0)), (1.0, (300, -200, 400)), (1.5, (0, 0, 0)), (1.5, (300, -200, 400)), (2.0, (0, 0, 0)), (2.0, (300, -200, 400))]:
    sp = abi.set_optics(sp0, blur, *sh)
    det.synth_render(sp, poses, frames)
    dets, fcs = det.detect(frames, n)
    img = det.fetch_images(n); lst = det.fetch_lists(n)
    found = {int(d.frame) for d in dets}
    a = b = c = 0; tot = 0; ncand = []; nkept = []; stat = []
    minscore = []
    for f in range(n):
        gt = synth.project_points(objb, poses[f][:3], poses[f][3:], K)
        cand = img["cand"][f][:min(img["cand_count"][f], cfg.max_candidates)]
        pre = lst["pre"][f][:lst["npre"][f]]
        kept = lst["kept"][f][:fcs[f].nkept]
        ncand.append(int(img["cand_count"][f])); nkept.append(int(fcs[f].nkept)); stat.append(int(fcs[f].status))
        for g in gt:
            tot += 1
            def near(L):
                if len(L) == 0: return False
                return bool(((np.abs(L["x"] - g[0]) <= 3) & (np.abs(L["y"] - g[1]) <= 3)).any())
            a += near(cand); b += near(pre); c += near(kept)
            if len(cand):
                m = (np.abs(cand["x"] - g[0]) <= 3) & (np.abs(cand["y"] - g[1]) <= 3)
                if m.any(): minscore.append(int(cand["score"][m].max()))
    print("blur %s shade %s: found %d/%d | GT corners with candidate %.3f, after suppression %.3f, validated %.3f | cand/frame med %d max %d, kept med %d, status!=0 %d | score at corners p1 %.0f p10 %.0f med %.0f" % (
        blur, sh, len(found), n, a / tot, b / tot, c / tot, np.median(ncand), max(ncand), np.median(nkept), sum(s != 0 for s in stat),
        np.percentile(minscore, 1) if minscore else -1, np.percentile(minscore, 10) if minscore else -1, np.median(minscore) if minscore else -1), flush=True)
