#!/usr/bin/env python3
"""experiment: do two batches in flight on two handles (own buffers, own streams) overlap usefully?  The tail of a batch
(list, sub-pixel, lattice + pose) is latency-bound with the GPU mostly idle; beside the next batch's head it could hide."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from robot_camera_calibration_amd import abi, api, synth
B = 1024
def mk():
    cfg = api.default_config(); abi.set_geometry(cfg, 1920, 1080); cfg.batch_capacity = B
    return cfg, api.Detector(cfg)
cfg, d0 = mk(); _, d1 = mk()
frames = torch.empty((B, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
sp = abi.default_synth_params(); poses = synth.sample_poses(B, cfg)
for s0 in range(0, B, 64):
    d0.synth_render(sp, poses[s0:s0 + 64], frames[s0:s0 + 64], first_index=s0)
torch.cuda.synchronize()
K = 20
def one_handle():
    d0.submit(frames, B)
    for k in range(K):
        if k + 1 < K: d0.submit(frames, B)
        r, _ = d0.collect()
    return len(r)
def two_handles(depth):
    # batch k goes to handle k & 1; `depth` submissions outstanding per handle
    dets = (d0, d1); pend = []
    n = 0
    for k in range(K):
        dets[k & 1].submit(frames, B); pend.append(k & 1)
        if len(pend) >= 2 * depth:
            r, _ = dets[pend.pop(0)].collect(); n = len(r)
    while pend:
        r, _ = dets[pend.pop(0)].collect(); n = len(r)
    return n
MODES = (("one handle, one batch ahead", one_handle), ("two handles, 1 batch each in flight", lambda: two_handles(1)), ("two handles, 2 each", lambda: two_handles(2)))
sel = int(sys.argv[1]) if len(sys.argv) > 1 else -1
for name, fn in (MODES if sel < 0 else MODES[sel:sel + 1]):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); n = fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / K * 1e3)
    print("%-40s %s ms per batch  (%d found)" % (name, ["%.3f" % t for t in ts], n))
