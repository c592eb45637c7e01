#!/usr/bin/env python3
"""Host-input / partial-batch fuzz: one handle of capacity B, random batch lengths n <= B, random chunking of the host pipeline
(rcc_set_host_chunk), pinned and pageable host memory, detect() and submit / collect in random interleavings, board and tag
scenes: every call must return exactly the records of the same frames as device-resident input.  usage: fuzz_hostpath.py SECONDS [SEED]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from robot_camera_calibration_amd import abi, api, synth
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time(); calls = 0; bad = 0; handles = 0
while time.time() - t0 < budget:
    tags = rng.random() < 0.35
    w, h = [(640, 480), (1280, 720), (960, 540)][int(rng.integers(3))]
    B = int(rng.integers(20, 120))
    pix = abi.RCC_PIX_BGR8 if rng.random() < 0.7 else abi.RCC_PIX_MONO8
    cfg = api.default_config(); abi.set_geometry(cfg, w, h, pix); cfg.batch_capacity = B
    sp = abi.default_synth_params(seed=int(rng.integers(1, 1 << 30)))
    if tags:
        abi.set_fiducial_target(cfg, abi.load_family(), tag_size=0.10)
        (hx, hy), _, _ = synth.fiducial_grid_layout(3, 2, cfg.tag_size)
        sp.fid_grid_x, sp.fid_grid_y, sp.fid_gap_permille = 3, 2, 500
    det = api.Detector(cfg); handles += 1
    poses = synth.sample_poses(B, cfg, seed=int(rng.integers(1, 1 << 30)), **(dict(z_range=(0.6, 1.4), max_tilt_deg=35, half_extent_m=(hx, hy)) if tags else {}))
    frames = torch.empty((B, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
    for s0 in range(0, B, 16):
        det.synth_render(sp, poses[s0:s0 + 16], frames[s0:min(s0 + 16, B)], first_index=s0)
    for z in rng.integers(0, B, 2): frames[int(z)].zero_()             # frames without a target
    torch.cuda.synchronize()
    pinned = torch.empty((B, cfg.frame_bytes), dtype=torch.uint8, pin_memory=True); pinned.copy_(frames); torch.cuda.synchronize()
    pageable = pinned.numpy().copy()
    ref = {}
    def want(n):
        if n not in ref:
            d, f = det.detect(frames[:n], n); ref[n] = (d.tobytes(), f.tobytes())
        return ref[n]
    for _ in range(12):
        n = int(rng.integers(1, B + 1))
        det.set_host_chunk(int(rng.choice([0, -1, 1, 3, 8, 17, 64])))
        src = [frames[:n], pinned[:n], pageable[:n]][int(rng.integers(3))]
        mode = int(rng.integers(3))
        exp = want(n)
        if mode == 0:
            d, f = det.detect(src, n); got = [(d.tobytes(), f.tobytes())]
        else:
            if mode == 2:
                n2 = int(rng.integers(1, B + 1)); exp2 = want(n2)          # (the synchronous call is refused while a submission is out)
            det.submit(src, n, want_corners=True)
            if mode == 2:
                src2 = [frames[:n2], pinned[:n2], pageable[:n2]][int(rng.integers(3))]
                det.submit(src2, n2, want_corners=True)
            d, f = det.collect(); got = [(d.tobytes(), f.tobytes())]
            if mode == 2:
                d, f = det.collect()
                if (d.tobytes(), f.tobytes()) != exp2: bad += 1; print("MISMATCH second of two submissions", w, h, B, n, n2, tags, flush=True)
        calls += 1
        if got[0] != exp:
            bad += 1; print("MISMATCH", dict(w=w, h=h, B=B, n=n, tags=tags, mode=mode, src=type(src).__name__), flush=True)
    det.close(); del frames, pinned; torch.cuda.empty_cache()
print("fuzz_hostpath: %d handles, %d calls, %d mismatches" % (handles, calls, bad))
sys.exit(1 if bad else 0)
