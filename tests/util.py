"""Shared helpers for the parity tests (test code only)."""
import ctypes as C

import numpy as np

from robot_camera_calibration_amd import abi


def clone_cfg(cfg):
    c = abi.rcc_config()
    C.memmove(C.byref(c), C.byref(cfg), C.sizeof(cfg))
    return c


def fc_px(fc, n):
    return np.array([[fc.px[k][0], fc.px[k][1]] for k in range(n)], np.int64)


def fc_xy(fc, n):
    return np.array([[fc.xy[k][0], fc.xy[k][1]] for k in range(n)], np.float64)


def sorted_cands(c):
    """candidate records sorted by (y, x) -- the dense pass's list is unordered by design"""
    c = np.asarray(c)
    order = np.lexsort((c["x"], c["y"]))
    return c[order]


def clutter_bgr(img, seed, count, keep_out=None, lo=8, hi=60):
    """paste `count` random rectangles of random colour over a BGR frame (h, w, 3), leaving the box keep_out = (x0, y0, x1, y1)
    alone: corners, edges and -- where rectangles overlap -- junction-like points all over the scene"""
    rng = np.random.default_rng(seed)
    out = img.copy()
    h, w = out.shape[:2]
    n = tries = 0
    while n < count and tries < 50 * count:
        tries += 1
        rw, rh = int(rng.integers(lo, hi)), int(rng.integers(lo, hi))
        x, y = int(rng.integers(0, w - rw)), int(rng.integers(0, h - rh))
        if keep_out is not None and x < keep_out[2] and x + rw > keep_out[0] and y < keep_out[3] and y + rh > keep_out[1]:
            continue
        out[y:y + rh, x:x + rw] = rng.integers(0, 256, 3)
        n += 1
    return out


def off_centre_board_among_clutter(oracle, cfg):
    """(frame h x w x 3, plain frame, ground-truth corners): a board in the lower right of a 1280x720 frame among 900 rectangles that
    keep 70 px from it -- the eight validated points nearest the centroid of all validated points are clutter, so the lattice stage
    finds the board only through its second seed group (the strongest points)"""
    import numpy as np
    from robot_camera_calibration_amd import abi, synth
    w, h = cfg.width, cfg.height
    pose = np.array([-0.2, 0.15, -0.3, 0.8, 0.38, 2.7])
    sp = abi.default_synth_params(seed=77)
    plain = np.asarray(oracle.synth_render(cfg, sp, pose, 1)).reshape(h, w, 3)
    gt = synth.project_points(synth.board_object_points(8, 6, 0.108), pose[:3], pose[3:], np.array(list(cfg.K)))
    ko = (gt[:, 0].min() - 70, gt[:, 1].min() - 70, gt[:, 0].max() + 70, gt[:, 1].max() + 70)
    return clutter_bgr(plain, 4910, 900, ko, 8, 50), plain, gt
