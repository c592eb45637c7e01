#!/usr/bin/env python3
"""Soak of the band kernel's counted wait: REPS launches of the stage form on noisy 4K and 1080p frames, every launch's
binary image and candidate set compared with the LDS kernel's (variant 0), one process.  usage: t_soak.py [REPS]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robot_camera_calibration_amd import abi, api, synth
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0
for (W, H, B) in [(3840, 2160, 48), (1920, 1080, 192), (2064, 1080, 64)]:
    cfg = api.default_config(); abi.set_geometry(cfg, W, H); cfg.batch_capacity = B
    det = api.Detector(cfg)
    frames = torch.empty((B, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
    sp = abi.default_synth_params(); poses = synth.sample_poses(16, cfg)
    poses = np.concatenate([poses] * ((B + 15) // 16))[:B]
    for s0 in range(0, B, 16):
        det.synth_render(sp, poses[s0:s0 + 16], frames[s0:s0 + 16], first_index=s0)
    px = W * H
    grey = torch.empty((B, px), dtype=torch.uint8, device="cuda:0")
    det.stage_ingest(frames, B, grey); torch.cuda.synchronize()
    # noise on top: more active rows, fewer skipped units
    g = torch.Generator(device="cuda:0"); g.manual_seed(7)
    grey[: B // 2] = (grey[: B // 2].to(torch.int16) + torch.randint(-12, 13, (B // 2, px), device="cuda:0", generator=g, dtype=torch.int16)).clamp(0, 255).to(torch.uint8)
    torch.cuda.synchronize()       # the stage calls run on the handle's stream: torch's stream must be idle first
    ref_bin = torch.empty_like(grey); ref_cand = torch.empty((B, cfg.max_candidates * 8), dtype=torch.uint8, device="cuda:0"); ref_cnt = torch.empty((B,), dtype=torch.int32, device="cuda:0")
    det.set_dense_variant(0)
    det.stage_threshold_corner(grey, B, ref_bin, ref_cand, ref_cnt); torch.cuda.synchronize()
    def cset(cand, cnt):
        c = cand.cpu().numpy().view(np.int16).reshape(B, -1, 4); n = cnt.cpu().numpy()
        return [set(map(tuple, c[f, : min(n[f], c.shape[1])].tolist())) for f in range(B)]
    ref_set = cset(ref_cand, ref_cnt)
    det.set_dense_variant(1)
    binm = torch.empty_like(grey); cand = torch.empty_like(ref_cand); cnt = torch.empty_like(ref_cnt)
    for r in range(REPS):
        binm.zero_(); torch.cuda.synchronize()
        det.stage_threshold_corner(grey, B, binm, cand, cnt); torch.cuda.synchronize()
        okb = bool(torch.equal(binm, ref_bin)); okc = bool(torch.equal(cnt, ref_cnt)) and cset(cand, cnt) == ref_set
        if not (okb and okc):
            bad += 1
            d = (binm != ref_bin).view(B, H, W)
            nd = int(d.sum().item())
            msg = "MISMATCH %dx%d rep %d: binary %s (%d px) candidates %s" % (W, H, r, okb, nd, okc)
            if nd:
                idx = torch.nonzero(d)
                f0, y0, x0 = idx[0].tolist()
                msg += "; frames %s first (f %d y %d x %d) ref %d got %d; y %d..%d x %d..%d" % (sorted(set(idx[:, 0].tolist()))[:8], f0, y0, x0, int(ref_bin.view(B, H, W)[f0, y0, x0]), int(binm.view(B, H, W)[f0, y0, x0]), int(idx[:, 1].min()), int(idx[:, 1].max()), int(idx[:, 2].min()), int(idx[:, 2].max()))
            if not okc:
                nb = cnt.cpu().numpy(); nr = ref_cnt.cpu().numpy()
                msg += "; count diffs at frames %s" % (np.nonzero(nb != nr)[0][:8].tolist(),)
            print(msg)
    print("%dx%d x %d: %d launches compared, kernel %s, mean candidates %.0f" % (W, H, B, REPS, det.last_dense_kernel(), ref_cnt.float().mean().item()))
    det.close(); del frames, grey, binm, cand, cnt, ref_bin, ref_cand, ref_cnt; torch.cuda.empty_cache()
print("soak:", "FAILED" if bad else "ok", bad)
sys.exit(1 if bad else 0)
