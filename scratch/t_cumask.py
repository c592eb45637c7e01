#!/usr/bin/env python3
"""experiment: two handles, each on a stream confined to a share of the CUs (hipExtStreamCreateWithCUMask), a batch each in
flight: does the idle time of one batch's tail get used by the other batch's head?"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from robot_camera_calibration_amd import abi, api, synth
hip = C.CDLL("libamdhip64.so")
def masked_stream(bits):
    words = (len(bits) + 31) // 32
    arr = (C.c_uint32 * words)()
    for i, b in enumerate(bits):
        if b: arr[i // 32] |= 1 << (i % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), C.c_uint32(words), arr)
    assert rc == 0, rc
    return s
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
NCU = torch.cuda.get_device_properties(0).multi_processor_count
K = 20
def run(sa, sb):
    dets = (d0, d1); strs = (sa, sb); pend = []
    n = 0
    for k in range(K):
        dets[k & 1].submit(frames, B, stream=strs[k & 1].value if strs[k & 1] is not None else None); pend.append(k & 1)
        if len(pend) >= 2:
            r, _ = dets[pend.pop(0)].collect(); n = len(r)
    while pend:
        r, _ = dets[pend.pop(0)].collect(); n = len(r)
    return n
def timeit(name, sa, sb):
    run(sa, sb); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); n = run(sa, sb); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / K * 1e3)
    print("%-52s %s ms per batch (%d found)" % (name, ["%.3f" % t for t in ts], n))
timeit("two handles, own unmasked streams", None, None)
full = [1] * NCU
timeit("two handles, both streams masked to ALL CUs", masked_stream(full), masked_stream(full))
lo = [1 if i < NCU // 2 else 0 for i in range(NCU)]; hi = [1 - b for b in lo]
timeit("halves: low / high CU indices", masked_stream(lo), masked_stream(hi))
ev = [1 if (i % 2 == 0) else 0 for i in range(NCU)]; od = [1 - b for b in ev]
timeit("halves: even / odd CU indices", masked_stream(ev), masked_stream(od))
q3 = [1 if (i % 8) < 6 else 0 for i in range(NCU)]; q1 = [1 - b for b in q3]
timeit("three quarters / one quarter (i % 8 < 6)", masked_stream(q3), masked_stream(q1))
