#!/usr/bin/env python3
"""End-to-end determinism soak: REPS x rcc_detect_batch and REPS x submit/collect on the same frames; every result record
must equal the first run's bit for bit (the compact-map form of the band kernel, list, sub-pixel, grid + pose)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robot_camera_calibration_amd import abi, api, synth
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0
for (W, H, B) in [(1920, 1080, 256), (3840, 2160, 48)]:
    cfg = api.default_config(); abi.set_geometry(cfg, W, H); cfg.batch_capacity = B
    det = api.Detector(cfg)
    frames = torch.empty((B, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
    sp = abi.default_synth_params(); poses = synth.sample_poses(B, cfg)
    for s0 in range(0, B, 16):
        det.synth_render(sp, poses[s0:s0 + 16], frames[s0:s0 + 16], first_index=s0)
    torch.cuda.synchronize()
    d0, f0 = det.detect(frames, B)
    ref = (d0.tobytes(), f0.tobytes())
    for r in range(REPS):
        d, f = det.detect(frames, B)
        if (d.tobytes(), f.tobytes()) != ref:
            bad += 1; print("MISMATCH detect %dx%d rep %d" % (W, H, r))
    det.submit(frames, B, want_corners=True)
    for r in range(REPS):
        det.submit(frames, B, want_corners=True)
        d, f = det.collect()
        if (d.tobytes(), f.tobytes()) != ref:
            bad += 1; print("MISMATCH stream %dx%d rep %d" % (W, H, r))
    det.collect()
    print("%dx%d x %d: %d + %d runs, %d targets, all identical: %s" % (W, H, B, REPS, REPS, len(d0), bad == 0))
    det.close(); del frames; torch.cuda.empty_cache()
# tag scenes: 256 frames x 24 tags (the sub-pixel kernel's narrow grid, the two-part segment test, refine_edges corners)
for refine in (abi.RCC_TAG_REFINE_EDGES, abi.RCC_TAG_REFINE_CORNER_SUBPIX):
    B = 256
    cfg = api.default_config(); abi.set_geometry(cfg, 1920, 1080); cfg.batch_capacity = B
    abi.set_fiducial_target(cfg, abi.load_family(), tag_size=0.10); cfg.tag_refine = refine
    det = api.Detector(cfg)
    frames = torch.empty((B, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
    sp = abi.default_synth_params()
    (fhx, fhy), _, _ = synth.fiducial_grid_layout(6, 4, cfg.tag_size)
    sp.fid_grid_x, sp.fid_grid_y, sp.fid_gap_permille = 6, 4, 500
    poses = synth.sample_poses(B, cfg, z_range=(1.0, 2.0), max_tilt_deg=40, half_extent_m=(fhx, fhy))
    for s0 in range(0, B, 16):
        det.synth_render(sp, poses[s0:s0 + 16], frames[s0:s0 + 16], first_index=s0)
    torch.cuda.synchronize()
    d0, f0 = det.detect(frames, B)
    ref = (d0.tobytes(), f0.tobytes())
    for r in range(REPS):
        d, f = det.detect(frames, B)
        if (d.tobytes(), f.tobytes()) != ref:
            bad += 1; print("MISMATCH tags refine %d rep %d" % (refine, r))
    for width in (16, 300, 0):
        det.set_subpix_grid(width)
        d, f = det.detect(frames, B)
        if (d.tobytes(), f.tobytes()) != ref:
            bad += 1; print("MISMATCH tags refine %d grid width %d" % (refine, width))
    print("tags, refine %d: %d runs + 3 grid widths, %d tags of %d, all identical: %s" % (refine, REPS, len(d0), B * 24, bad == 0))
    det.close(); del frames; torch.cuda.empty_cache()
print("soak e2e:", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
