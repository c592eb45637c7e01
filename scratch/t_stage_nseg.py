#!/usr/bin/env python3
"""experiment (librcc_hip_exp.so, RCC_DENSE_NSEG): stage form of the threshold + corner pass by segments per frame"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    os.environ["RCC_LIBRARY"] = os.path.join(ROOT, "robot_camera_calibration_amd", "librcc_hip_exp.so")
    import numpy as np, torch
    from robot_camera_calibration_amd import abi, api, synth
    B = 1024
    cfg = api.default_config(); abi.set_geometry(cfg, 1920, 1080); cfg.batch_capacity = B
    det = api.Detector(cfg)
    frames = torch.empty((B, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
    sp = abi.default_synth_params(); poses = synth.sample_poses(B, cfg)
    for s0 in range(0, B, 64):
        det.synth_render(sp, poses[s0:s0 + 64], frames[s0:s0 + 64], first_index=s0)
    px = 1920 * 1080
    grey = torch.empty((B, px), dtype=torch.uint8, device="cuda:0"); binm = torch.empty_like(grey)
    cand = torch.empty((B, cfg.max_candidates * 8), dtype=torch.uint8, device="cuda:0"); cnt = torch.empty((B,), dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    det.stage_ingest(frames, B, grey)
    det.time_dense(grey, B, binm, cand, cnt, 1)
    ms = [det.time_dense(grey, B, binm, cand, cnt, 5) for _ in range(3)]
    print("nseg", os.environ.get("RCC_DENSE_NSEG", "auto"), ["%.3f" % m for m in ms], det.last_dense_kernel())
else:
    for n in ("", "2", "3", "4", "5", "6", "8", "9", "10", "12"):
        env = dict(os.environ)
        if n: env["RCC_DENSE_NSEG"] = n
        subprocess.run([sys.executable, os.path.abspath(__file__), "run"], env=env)
