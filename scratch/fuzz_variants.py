#!/usr/bin/env python3
"""Variant / layout fuzz: the same frames through a handle with default settings and tight rows, and through a second handle with a
random combination of the switches between bit-identical variants (dense variant, skip, gang, ingest variant, fused lattice + pose,
pipeline chunks, pose mapping, kept binary image, sub-pixel grid width) and a padded memory layout (row stride and frame pitch
beyond the pixels, sometimes not multiples of 16: the fall-back kernels): the records and corner tables must be identical.
usage: fuzz_variants.py SECONDS [SEED]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
EXP = os.path.basename(os.environ.get('RCC_LIBRARY', '')) == 'librcc_hip_exp.so'     # the gang form (and a real variant 3) exist only there
from robot_camera_calibration_amd import abi, api, synth
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time(); runs = 0; bad = 0
while time.time() - t0 < budget:
    tags = rng.random() < 0.3
    w, h = [(640, 480), (1280, 720), (1920, 1080), (960, 540), (2064, 1544), (645, 483)][int(rng.integers(6))]
    if tags and w % 16: w, h = 1280, 720
    pix = abi.RCC_PIX_BGR8 if rng.random() < 0.7 else abi.RCC_PIX_MONO8
    ch = 3 if pix == abi.RCC_PIX_BGR8 else 1
    n = int(rng.integers(2, 20))
    model = int(rng.integers(0, 3))
    def make(stride, fbytes):
        cfg = api.default_config(); abi.set_geometry(cfg, w, h, pix); cfg.batch_capacity = n
        cfg.stride_bytes = stride; cfg.frame_bytes = fbytes
        cfg.dist_model = model
        for i in range(8): cfg.D[i] = 0.0
        if model == abi.RCC_DIST_PLUMB_BOB:
            for i, v in enumerate(abi.PLUMB_BOB_DEFAULT): cfg.D[i] = v
        elif model == abi.RCC_DIST_FISHEYE:
            for i, v in enumerate((-0.05, 0.01, -0.002, 0.0003)): cfg.D[i] = v
        cfg.undistort = 1 if model == abi.RCC_DIST_FISHEYE else und
        if tags:
            abi.set_fiducial_target(cfg, abi.load_family(), tag_size=0.10)
        return cfg
    und = int(rng.random() < 0.7)
    tight = make(w * ch, w * ch * h)
    pad_row = int(rng.choice([0, 16, 48, 5, 1, 64]))
    pad_frame = int(rng.choice([0, 16, 256, 7, 4096]))
    stride = w * ch + pad_row
    padded = make(stride, stride * h + pad_frame)
    desc = dict(w=w, h=h, pix=pix, n=n, model=model, und=und, tags=tags, pad_row=pad_row, pad_frame=pad_frame)
    try:
        d0 = api.Detector(tight)
        sp = abi.default_synth_params(seed=int(rng.integers(1, 1 << 30)))
        kw = {}
        if tags:
            (hx, hy), _, _ = synth.fiducial_grid_layout(3, 2, tight.tag_size)
            sp.fid_grid_x, sp.fid_grid_y, sp.fid_gap_permille = 3, 2, 500
            kw = dict(z_range=(0.6, 1.4), max_tilt_deg=35, half_extent_m=(hx, hy))
        poses = synth.sample_poses(n, tight, seed=int(rng.integers(1, 1 << 30)), **kw)
        f0 = torch.empty((n, tight.frame_bytes), dtype=torch.uint8, device="cuda:0")
        d0.synth_render(sp, poses, f0)
        torch.cuda.synchronize()
        ref_d, ref_f = d0.detect(f0, n)
        d0.close()
        # the same pixels in the padded layout (padding bytes random)
        f1 = torch.randint(0, 256, (n, padded.frame_bytes), dtype=torch.uint8, device="cuda:0")
        f1[:, :stride * h].view(n, h, stride)[:, :, :w * ch] = f0.view(n, h, w * ch)
        d1 = api.Detector(padded)
        sw = dict(dense=int(rng.choice([-1, 0, 1, 2, 3, 4])), skip=int(rng.integers(2)), gang=int(rng.choice([0, 0, 1, 4, 16])) if EXP else 0, ingest=int(rng.choice([-1, 0, 1, 2, 3] if EXP else [-1, 0, 1, 2])),
                  fuse=int(rng.integers(2)), pipe=int(rng.choice([0, 0, 2, 3])), pnp=int(rng.choice([-1, 1])), keep=int(rng.integers(2)), grid=int(rng.choice([0, 1, 5, 64])))
        d1.set_dense_variant(sw["dense"]); d1.set_dense_skip(sw["skip"]); d1.set_ingest_variant(sw["ingest"])
        if EXP: d1.set_dense_gang(sw["gang"])
        d1.set_fuse_grid_pnp(sw["fuse"]); d1.set_pipeline(sw["pipe"]); d1.set_pnp_variant(sw["pnp"]); d1.set_keep_binary(sw["keep"]); d1.set_subpix_grid(sw["grid"])
        desc.update(sw)
        for rep in range(2):
            d, f = d1.detect(f1, n)
            if d.tobytes() != ref_d.tobytes() or f.tobytes() != ref_f.tobytes():
                bad += 1; print("MISMATCH", desc, "rep", rep, "records", len(d), len(ref_d), flush=True)
                if bad <= 12:
                    for name in ref_d.dtype.names:
                        if len(d) == len(ref_d) and not np.array_equal(np.asarray(d[name]), np.asarray(ref_d[name])):
                            a, b = np.asarray(d[name], float), np.asarray(ref_d[name], float)
                            print("     records differ in", name, "max |diff| %.3g" % np.abs(a - b).max(), flush=True)
                    for name in ref_f.dtype.names:
                        if not np.array_equal(np.asarray(f[name]), np.asarray(ref_f[name])):
                            print("     corner tables differ in", name, flush=True)
                break
        d1.close(); del f0, f1; torch.cuda.empty_cache()
    except Exception as e:
        bad += 1; print("ERROR", desc, repr(e)[:300], flush=True)
    runs += 1
    if runs % 50 == 0: print("%d runs, %d bad, %.0f s" % (runs, bad, time.time() - t0), flush=True)
print("fuzz_variants: %d runs, %d mismatches / errors" % (runs, bad))
sys.exit(1 if bad else 0)
