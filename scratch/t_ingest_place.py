#!/usr/bin/env python3
"""Does the ingest pass's speed depend on where its destination lies relative to the source?  (t_clock.py saw 1.65 vs 1.8 ms
for two destination buffers.)  Times the pass alone, 6 launches each, for a range of destination offsets."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robot_camera_calibration_amd import abi, api, synth
B = 1024
cfg = api.default_config(); abi.set_geometry(cfg, 1920, 1080); cfg.batch_capacity = B
det = api.Detector(cfg)
frames = torch.empty((B, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
sp = abi.default_synth_params(); poses = synth.sample_poses(32, cfg)
poses = np.concatenate([poses] * ((B + 31) // 32))[:B]
for s0 in range(0, B, 64):
    det.synth_render(sp, poses[s0:s0 + 64], frames[s0:s0 + 64], first_index=s0)
px = 1920 * 1080
big = torch.empty((B * px + (64 << 20),), dtype=torch.uint8, device="cuda:0")
other = torch.empty((B * px,), dtype=torch.uint8, device="cuda:0")
st = torch.cuda.current_stream(); sh = st.cuda_stream
torch.cuda.synchronize()
print("frames at 0x%x, big at 0x%x, other at 0x%x" % (frames.data_ptr(), big.data_ptr(), other.data_ptr()))

def t(dst, reps=6):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record(st)
    for r in range(reps):
        det.stage_ingest(frames, B, dst, stream=sh); ev[r + 1].record(st)
    torch.cuda.synchronize()
    return [ev[r].elapsed_time(ev[r + 1]) for r in range(reps)]

for rnd in range(2):
    for off in [0, 256, 1024, 4096, 16384, 65536, 1 << 18, 1 << 20, 1 << 22, 1 << 24, (1 << 24) + 4096, 1 << 25]:
        ts = t(big[off:off + B * px])
        print("offset %9d: " % off + " ".join("%.3f" % x for x in ts))
    print("other buffer:     " + " ".join("%.3f" % x for x in t(other)))
