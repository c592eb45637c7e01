#!/usr/bin/env python3
"""Stage-call fuzz: rcc_stage_ingest -> rcc_stage_threshold_corner -> rcc_stage_targets over the caller's own buffers (full binary
image) against rcc_detect_batch (compact threshold map) on the same frames: identical records and frame tables.
usage: fuzz_stages.py SECONDS [SEED]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from robot_camera_calibration_amd import abi, api, synth
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time(); runs = 0; bad = 0
while time.time() - t0 < budget:
    tags = rng.random() < 0.3
    w, h = [(640, 480), (1280, 720), (1920, 1080), (960, 540), (645, 483), (322, 241), (2064, 1544)][int(rng.integers(7))]
    if tags and w % 16: w, h = 1280, 720
    pix = abi.RCC_PIX_BGR8 if rng.random() < 0.7 else abi.RCC_PIX_MONO8
    n = int(rng.integers(1, 24))
    cfg = api.default_config(); abi.set_geometry(cfg, w, h, pix); cfg.batch_capacity = n + int(rng.integers(0, 5))
    model = int(rng.integers(0, 3)); cfg.dist_model = model
    for i in range(8): cfg.D[i] = 0.0
    if model == abi.RCC_DIST_PLUMB_BOB:
        for i, v in enumerate(abi.PLUMB_BOB_DEFAULT): cfg.D[i] = v
    elif model == abi.RCC_DIST_FISHEYE:
        for i, v in enumerate((-0.05, 0.01, -0.002, 0.0003)): cfg.D[i] = v
    cfg.undistort = 1 if model == abi.RCC_DIST_FISHEYE else int(rng.random() < 0.7)
    desc = dict(w=w, h=h, pix=pix, n=n, model=model, und=cfg.undistort, tags=tags)
    try:
        sp = abi.default_synth_params(seed=int(rng.integers(1, 1 << 30)))
        kw = {}
        if tags:
            abi.set_fiducial_target(cfg, abi.load_family(), tag_size=0.10)
            (hx, hy), _, _ = synth.fiducial_grid_layout(3, 2, cfg.tag_size)
            sp.fid_grid_x, sp.fid_grid_y, sp.fid_gap_permille = 3, 2, 500
            kw = dict(z_range=(0.6, 1.4), max_tilt_deg=35, half_extent_m=(hx, hy))
        det = api.Detector(cfg)
        poses = synth.sample_poses(n, cfg, seed=int(rng.integers(1, 1 << 30)), **kw)
        frames = torch.empty((n, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
        det.synth_render(sp, poses, frames)
        if n > 2: frames[int(rng.integers(n))].zero_()
        torch.cuda.synchronize()
        d, f = det.detect(frames, n)
        px = w * h
        grey = torch.empty((n, px), dtype=torch.uint8, device="cuda:0"); binm = torch.empty((n, px), dtype=torch.uint8, device="cuda:0")
        cand = torch.empty((n, cfg.max_candidates * 8), dtype=torch.uint8, device="cuda:0"); cnt = torch.zeros((n,), dtype=torch.int32, device="cuda:0")
        det.stage_ingest(frames, n, grey)
        det.stage_threshold_corner(grey, n, binm, cand, cnt)
        d2, f2 = det.stage_targets(grey, binm, cand, cnt, n)
        if d.tobytes() != d2.tobytes() or f.tobytes() != f2.tobytes():
            bad += 1; print("MISMATCH stage calls vs detect", desc, len(d), len(d2), flush=True)
        det.close(); del frames, grey, binm, cand; torch.cuda.empty_cache()
    except Exception as e:
        bad += 1; print("ERROR", desc, repr(e)[:300], flush=True)
    runs += 1
print("fuzz_stages: %d runs, %d mismatches / errors" % (runs, bad))
sys.exit(1 if bad else 0)
