#!/usr/bin/env python3
"""How much of the bench frames is 'active' for the corner stages of the threshold+corner pass, as a function of the width
of the unit a wavefront would skip at -- the numbers behind DESIGN.md section 5 (why neither narrower windows nor a
compacted work list rescue the two-kernel form).  CPU only: frames rendered and ingested by the oracle (slow: ~8 s per
1080p frame), tile statistics in numpy.  usage: activity_cpu.py [nframes=24]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robot_camera_calibration_amd import abi, synth
from oracle import orc_py
W, H = 1920, 1080
N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
cfg = orc_py.default_config(); abi.set_geometry(cfg, W, H)
sp = abi.default_synth_params()
poses = synth.sample_poses(N, cfg)
g = np.stack([orc_py.ingest(cfg, orc_py.synth_render(cfg, sp, poses[f], f)) for f in range(N)])
mc = cfg.thr_min_contrast
t = g.reshape(N, H // 4, 4, W // 4, 4).transpose(0, 1, 3, 2, 4).reshape(N, H // 4, W // 4, 16).astype(np.int16)
tmin, tmax = t.min(-1), t.max(-1)
def dil(a, f):
    p = np.pad(a, ((0, 0), (1, 1), (1, 1)), mode="edge"); r = a.copy()
    for dy in range(3):
        for dx in range(3): r = f(r, p[:, dy:dy + a.shape[1], dx:dx + a.shape[2]])
    return r
act = (dil(tmax, np.maximum) - dil(tmin, np.minimum)) >= mc          # non-flat tile: the skip rule's own test
print("tiles with contrast >= %d: %.4f; non-flat after the 3x3 dilation: %.4f" % (mc, ((tmax - tmin) >= mc).mean(), act.mean()))
TH, TW = act.shape[1:]
for uw in (1, 4, 13, 16, 29, 61):          # useful tiles per unit; a unit of n lanes has 3 halo lanes (12 px)
    nw = (TW + uw - 1) // uw
    u = np.pad(act, ((0, 0), (0, 0), (0, nw * uw - TW))).reshape(N, TH, nw, uw).any(-1)
    p = np.pad(u, ((0, 0), (1, 1), (0, 0)))
    uv = p[:, :-2] | p[:, 1:-1] | p[:, 2:]                             # + the tile rows above and below (warm-up / cool-down)
    print("unit %3d tiles (%4d px): active %.3f, with the rows above / below %.3f, lanes spent on halo %.0f %%" % (uw, 4 * uw, u.mean(), uv.mean(), 100.0 * 3 / uw))
