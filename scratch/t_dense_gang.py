#!/usr/bin/env python3
"""experiment: k_dense_wave as gangs of eight windows meeting every n tile rows (rcc_set_dense_gang): time and equality of outputs"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from robot_camera_calibration_amd import abi, api, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
only = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]]
cfg = api.default_config(); abi.set_geometry(cfg, 1920, 1080); cfg.batch_capacity = B
det = api.Detector(cfg)
frames = torch.empty((B, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
sp = abi.default_synth_params(); poses = synth.sample_poses(B, cfg)
for s0 in range(0, B, 64):
    det.synth_render(sp, poses[s0:s0 + 64], frames[s0:s0 + 64], first_index=s0)
px = 1920 * 1080
grey = torch.empty((B, px), dtype=torch.uint8, device="cuda:0")
cand = torch.empty((B, cfg.max_candidates * 8), dtype=torch.uint8, device="cuda:0"); cnt = torch.empty((B,), dtype=torch.int32, device="cuda:0")
torch.cuda.synchronize()
det.stage_ingest(frames, B, grey)
ref = None
combos = only or [(0, 0), (1, 0), (2, 0), (4, 0), (8, 0), (16, 0), (4, 4), (8, 4), (4, 3), (8, 3), (16, 3), (8, 2), (4, 6)]
for sync, seg in combos:
    det.set_dense_gang(sync, seg)
    det.time_dense(grey, B, None, cand, cnt, 1)
    ms = [det.time_dense(grey, B, None, cand, cnt, 5) for _ in range(3)]
    k = cnt.cpu().numpy().copy()
    c = cand.cpu().numpy().view(api.CAND_DT).reshape(B, cfg.max_candidates)
    sig = (k.sum(), sum(int(np.sort(c[f][:k[f]].view(np.int64)).sum() % (1 << 61)) for f in range(0, B, 37)))
    if ref is None: ref = sig
    print("gang sync %2d seg %d: %s ms  kernel %s  %s" % (sync, seg, ["%.3f" % m for m in ms], det.last_dense_kernel(), "same" if sig == ref else "DIFFERENT %s %s" % (sig, ref)))
