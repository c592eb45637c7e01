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
