#!/usr/bin/env python3
"""Where does the one-off stall of the streamed steps with corner tables sit: in submit or in collect, and on which step?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from robot_camera_calibration_amd import abi, api, synth
B = 1024
cfg = api.default_config(); abi.set_geometry(cfg, 1920, 1080, abi.RCC_PIX_BGR8); cfg.batch_capacity = B
det = api.Detector(cfg)
sp = abi.default_synth_params(); poses = synth.sample_poses(64, cfg); poses = np.concatenate([poses] * 16)
frames = torch.empty((B, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
for s0 in range(0, B, 64): det.synth_render(sp, poses[s0:s0 + 64], frames[s0:s0 + 64], first_index=s0)
torch.cuda.synchronize()
mode = sys.argv[1] if len(sys.argv) > 1 else "zeros"
if mode == "reuse":
    pool = [np.zeros(B, api.FC_DT) for _ in range(3)]
for rep in range(2):
    ts = []
    det.submit(frames, B, want_corners=True)
    for k in range(14):
        t0 = time.perf_counter(); det.submit(frames, B, want_corners=True); t1 = time.perf_counter()
        d, f = det.collect(); t2 = time.perf_counter()
        ts.append((round(1e3 * (t1 - t0), 2), round(1e3 * (t2 - t1), 2)))
    det.collect()
    print(mode, "rep", rep, "(submit ms, collect ms):", ts, flush=True)
