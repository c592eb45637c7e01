#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the CPU oracle.

The reference (Virtana/robot_camera_calibration) holds no fixtures for this path and cannot be
run here, so these vectors come from the build's own oracle ("parity unpinned");
they freeze its behaviour so that later rounds notice any drift, and give the GPU tests inputs
that do not depend on a renderer.  Data only: inputs + expected outputs.

  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc_py as O                                    # noqa: E402
from robot_camera_calibration_amd import abi, synth               # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def golden_cfg(w=320, h=240):
    cfg = O.default_config()
    abi.set_geometry(cfg, w, h, abi.RCC_PIX_BGR8)
    cfg.batch_capacity = 4
    return cfg


def main():
    cfg = golden_cfg()
    sp = abi.default_synth_params(seed=0xC0FFEE)
    poses = synth.sample_poses(2, cfg, seed=0xC0FFEE, z_range=(1.0, 1.4), max_tilt_deg=35)
    ctx = O.Context(cfg)
    out = dict(poses=poses)
    frames = []
    for f in range(2):
        img = O.synth_render(cfg, sp, poses[f], f)
        n, det, fc, st = ctx.detect(img, f, stages=True)
        assert n == 1, "golden frame %d: board not found" % f
        frames.append(img)
        out["grey_sha%d" % f] = np.frombuffer(hashlib.sha256(st["grey"].tobytes()).digest(), np.uint8)
        out["bin_sha%d" % f] = np.frombuffer(hashlib.sha256(st["bin"].tobytes()).digest(), np.uint8)
        out["cand%d" % f] = st["cand"]
        out["pre%d" % f] = st["pre"]
        out["pre_xy%d" % f] = st["pre_xy"]
        out["kept%d" % f] = st["kept"]
        out["px%d" % f] = np.array([[fc.px[k][0], fc.px[k][1]] for k in range(48)], np.int32)
        out["xy%d" % f] = np.array([[fc.xy[k][0], fc.xy[k][1]] for k in range(48)])
        out["rvec%d" % f] = np.array(det.rvec[:])
        out["tvec%d" % f] = np.array(det.tvec[:])
        out["rms%d" % f] = np.array([det.rms])
    out["frames"] = np.stack(frames)
    np.savez_compressed(os.path.join(HERE, "board_320x240.npz"), **out)

    # solvePnP golden: the reference's call shape (camera_pose.cpp:152-163): 4 corners bl,br,tr,tl,
    # object points (+-s/2, +-s/2, 0), K, D(5), int-truncated pixels in half of the cases
    rng = np.random.default_rng(2024)
    K = np.array(list(cfg.K)); D = np.array(list(cfg.D))
    objs, imgs, rv, tv, rms = [], [], [], [], []
    for t in range(64):
        s = rng.uniform(0.03, 0.1)
        obj = np.array([[-s, -s, 0], [s, -s, 0], [s, s, 0], [-s, s, 0]], float)
        R = synth.rodrigues([0, 0, rng.uniform(-3, 3)]) @ synth.rodrigues(np.array([np.cos(t), np.sin(t), 0]) * rng.uniform(0, 1.0)) @ np.diag([1., -1, -1])
        r0 = synth.rotmat_to_rvec(R); t0 = np.array([rng.uniform(-.2, .2), rng.uniform(-.15, .15), rng.uniform(0.5, 2.0)])
        img = synth.project_points(obj, r0, t0, K, abi.RCC_DIST_PLUMB_BOB, D)
        if t % 2:
            img = np.floor(img)
        st, r, tt, e, it = O.solve_pnp(obj, img, K, abi.RCC_DIST_PLUMB_BOB, D)
        assert st == 0
        objs.append(obj); imgs.append(img); rv.append(r); tv.append(tt); rms.append(e)
    np.savez_compressed(os.path.join(HERE, "pnp_tags.npz"), K=K, D=D, obj=np.array(objs), img=np.array(imgs),
                        rvec=np.array(rv), tvec=np.array(tv), rms=np.array(rms))
    for fn in ("board_320x240.npz", "pnp_tags.npz"):
        print(fn, os.path.getsize(os.path.join(HERE, fn)), "bytes")


if __name__ == "__main__":
    main()
