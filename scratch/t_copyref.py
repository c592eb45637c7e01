#!/usr/bin/env python3
"""the library's calibration copy against torch's elementwise 1:1 kernel, same 2.1 GB"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robot_camera_calibration_amd import abi, api
cfg = api.default_config(); abi.set_geometry(cfg, 1920, 1080); cfg.batch_capacity = 8
det = api.Detector(cfg)
n = 1920 * 1080 * 1024
a = torch.randint(0, 255, (n,), dtype=torch.uint8, device="cuda:0"); out = torch.empty_like(a)
torch.cuda.synchronize()
print("rcc_time_copy: %.3f ms" % det.time_copy(a, out, n, reps=8))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.add(a, 1, out=out); torch.cuda.synchronize(); e0.record()
for _ in range(8): torch.add(a, 1, out=out)
e1.record(); torch.cuda.synchronize(); print("torch add1: %.3f ms" % (e0.elapsed_time(e1) / 8))
assert bool((out == a + 1).all())
det.time_copy(a, out, n, reps=1); torch.cuda.synchronize(); assert bool((out == a).all()); print("copy correct")
