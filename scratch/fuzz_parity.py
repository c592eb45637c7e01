#!/usr/bin/env python3
"""Parity fuzz: random configurations (size, pixel format, distortion model and coefficients, noise, contrast, board pose,
board or tag scene, tag refinement) through rcc_detect_batch and the CPU oracle, every stage of every frame compared
(tests/test_gpu_parity.py::_check_batch; tag scenes: refined positions bit for bit, ids / corners / poses).  Not a test of the
suite: a longer hunt for rare disagreements.  usage: fuzz_parity.py SECONDS [SEED]"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from robot_camera_calibration_amd import abi, api, synth
from oracle import orc_py as oracle
from tests import test_gpu_parity as T

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
SIZES = [(640, 480), (1280, 720), (1920, 1080), (960, 540), (1024, 768), (645, 483), (2064, 1544), (2560, 1440)]
t0 = time.time(); runs = 0; fails = 0; frames_total = 0; found_total = 0
while time.time() - t0 < budget:
    w, h = SIZES[int(rng.integers(len(SIZES)))]
    pix = abi.RCC_PIX_BGR8 if rng.random() < 0.7 else abi.RCC_PIX_MONO8
    model = int(rng.integers(0, 3))
    n = int(rng.integers(2, 7))
    seed = int(rng.integers(1, 1 << 30))
    tags = rng.random() < float(os.environ.get("FUZZ_TAGS", "0.3")) and (w % 16 == 0)
    desc = dict(w=w, h=h, pix=pix, model=model, n=n, seed=seed, tags=tags)
    try:
        cfg = api.default_config()
        abi.set_geometry(cfg, w, h, pix)
        cfg.batch_capacity = n
        cfg.dist_model = model
        for i in range(8): cfg.D[i] = 0.0
        if model == abi.RCC_DIST_PLUMB_BOB:
            d = [rng.uniform(-0.35, 0.1), rng.uniform(-0.1, 0.15), rng.uniform(-2e-3, 2e-3), rng.uniform(-2e-3, 2e-3), rng.uniform(-0.05, 0.05)]
            for i, v in enumerate(d): cfg.D[i] = float(v)
        elif model == abi.RCC_DIST_FISHEYE:
            d = [rng.uniform(-0.1, 0.05), rng.uniform(-0.02, 0.02), rng.uniform(-5e-3, 5e-3), rng.uniform(-1e-3, 1e-3)]
            for i, v in enumerate(d): cfg.D[i] = float(v)
        cfg.undistort = 1 if model == abi.RCC_DIST_FISHEYE else int(rng.random() < 0.7)      # fisheye without undistortion: rcc_create refuses it (RCC_ERR_UNSUPPORTED)
        cfg.reference_mode = int(rng.random() < 0.2)
        cfg.thr_min_contrast, cfg.harris_thresh = [(5, 10240), (16, 10240), (32, 200000), (12, 3200), (16, 200000)][int(rng.integers(5))]
        desc.update(undistort=cfg.undistort, refmode=cfg.reference_mode, mc=cfg.thr_min_contrast, ht=cfg.harris_thresh, D=[round(cfg.D[i], 4) for i in range(5)])
        sp = abi.default_synth_params(seed=seed, noise=float(rng.choice([0.0, 1.0, 2.0, 5.0])))
        lo = int(rng.integers(10, 80)); sp.black, sp.white = lo, int(rng.integers(lo + 60, 250))
        # the camera's optics (rcc_synth_params, ABI 2): half of the configurations see blur and / or shading
        if rng.random() < 0.5:
            blur = [None, "3tap", 0.7, 1.0, 1.5, 2.0, float(rng.uniform(0.4, 2.3))][int(rng.integers(7))]
            sh = (int(rng.integers(-400, 401)), int(rng.integers(-300, 301)), int(rng.integers(0, 601))) if rng.random() < 0.7 else (0, 0, 0)
            if abs(sh[0]) + abs(sh[1]) > 1000: sh = (sh[0] // 2, sh[1] // 2, sh[2])
            abi.set_optics(sp, blur, *sh)
            desc.update(blur=blur, shade=sh)
        if tags:
            refine = abi.RCC_TAG_REFINE_EDGES if rng.random() < 0.6 else abi.RCC_TAG_REFINE_CORNER_SUBPIX
            abi.set_fiducial_target(cfg, abi.load_family(), tag_size=0.10); cfg.tag_refine = refine
            gx, gy = int(rng.integers(2, 7)), int(rng.integers(2, 5))
            (hx, hy), centres, ids = synth.fiducial_grid_layout(gx, gy, cfg.tag_size)
            sp.fid_grid_x, sp.fid_grid_y, sp.fid_gap_permille = gx, gy, int(rng.choice([300, 500, 800]))
            desc.update(grid=(gx, gy), refine=refine)
            det = api.Detector(cfg)
            poses = synth.sample_poses(n, cfg, seed=seed, z_range=(0.7, 2.2), max_tilt_deg=45, half_extent_m=(hx, hy))
            frames = torch.empty((n, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
            det.synth_render(sp, poses, frames)
            dets, fcs = det.detect(frames, n)
            lst = det.fetch_lists(n); img = det.fetch_images(n)
            host = frames.cpu().numpy()
            ctx = oracle.Context(cfg)
            k0 = 0
            for f in range(n):
                m, odet, ofc, st = ctx.detect(host[f], f, stages=True)
                assert (img["grey"][f] == st["grey"]).all() and (img["bin"][f] == st["bin"]).all(), "image stages"
                assert lst["npre"][f] == st["npre"] and fcs[f].status == ofc.status, "list"
                if st["npre"]:
                    assert np.abs(lst["pre_xy"][f][:st["npre"]] - st["pre_xy"]).max() == 0.0, "refined positions"
                mine = dets[k0:k0 + m]
                assert len(mine) == m and (mine.frame == f).all(), "tag count %d vs %d" % (int((dets.frame == f).sum()), m)
                for k in range(m):
                    a, b = mine[k], odet[k]
                    assert a.id == b.id and a.hamming == b.hamming and a.pnp_status == b.pnp_status, "tag record"
                    assert np.abs(a.corners - np.array([[b.corners[q][0], b.corners[q][1]] for q in range(4)])).max() == 0.0, "tag corners"
                    assert np.abs(a.rvec - np.array(b.rvec[:])).max() <= 1e-4 and np.abs(a.tvec - np.array(b.tvec[:])).max() <= 1e-4, "tag pose"
                k0 += m; found_total += m
            assert k0 == len(dets)
            ctx.close(); det.close()
        else:
            if rng.random() < 0.3:          # boards other than 8 x 6: down to 3 x 3, beyond 64 corners, square ones
                cfg.board_cols, cfg.board_rows = int(rng.integers(3, 14)), int(rng.integers(3, 10))
                cfg.board_square = float(min(0.9 / (cfg.board_cols + 1), 0.65 / (cfg.board_rows + 1)))
                sp.board_cols, sp.board_rows, sp.board_square = cfg.board_cols, cfg.board_rows, cfg.board_square
                desc.update(board=(cfg.board_cols, cfg.board_rows))
            det = api.Detector(cfg)
            poses = synth.sample_poses(n, cfg, seed=seed)
            frames = torch.empty((n, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
            det.synth_render(sp, poses, frames)
            torch.cuda.synchronize()
            det.close()
            if pix == abi.RCC_PIX_BGR8 and rng.random() < 0.3:
                # other objects in view (round 4): rectangles of random colour all over the frames, sometimes over the board too -- the
                # long suppressed lists, the validated list's capacity, the second seed group and the overflow statuses
                from tests.util import clutter_bgr
                count, lo_, hi_ = [(50, 8, 60), (200, 8, 60), (600, 8, 50), (1500, 8, 40), (4000, 6, 14)][int(rng.integers(5))]
                cfg.max_candidates = int(rng.choice([2048, 4096]))
                host = frames.cpu().numpy().reshape(n, h, w, 3)
                Kc = np.array(list(cfg.K)); objb = synth.board_object_points(cfg.board_cols, cfg.board_rows, cfg.board_square)
                for f in range(n):
                    ko = None
                    if rng.random() < 0.7 and not cfg.undistort:
                        gt = synth.project_points(objb, poses[f][:3], poses[f][3:], Kc)
                        ko = (gt[:, 0].min() - 60, gt[:, 1].min() - 60, gt[:, 0].max() + 60, gt[:, 1].max() + 60)
                    elif rng.random() < 0.7:
                        ko = (w * 0.2, h * 0.2, w * 0.8, h * 0.8)
                    host[f] = clutter_bgr(host[f], seed + f, count, ko, lo_, hi_)
                frames = torch.from_numpy(np.ascontiguousarray(host).reshape(n, -1)).cuda()
                desc.update(clutter=count, maxc=cfg.max_candidates)
            mx, found = T._check_batch(torch, oracle, cfg, frames, n, expect_found=False)
            found_total += found
        frames_total += n
    except Exception as e:
        fails += 1
        print("FAIL", desc, "->", repr(e)[:300], flush=True)
        traceback.print_exc(limit=2)
    runs += 1
    if runs % 20 == 0:
        print("%d configurations, %d frames, %d targets found, %d failures, %.0f s" % (runs, frames_total, found_total, fails, time.time() - t0), flush=True)
print("fuzz: %d configurations, %d frames, %d targets found, %d failures" % (runs, frames_total, found_total, fails))
sys.exit(1 if fails else 0)
