#!/usr/bin/env python3
"""Does the threshold+corner pass run slower right after the (bandwidth-bound) ingest pass, and is it the clock?
A compute-bound probe (fp64 2048^3 matmul) is timed after idle / after ingest / after the dense pass."""
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
grey = torch.empty((B, px), dtype=torch.uint8, device="cuda:0"); binm = torch.empty_like(grey)
grey2 = torch.empty_like(grey)
cand = torch.empty((B, cfg.max_candidates * 8), dtype=torch.uint8, device="cuda:0"); cnt = torch.empty((B,), dtype=torch.int32, device="cuda:0")
A = torch.randn(2048, 2048, dtype=torch.float64, device="cuda:0"); Bm = torch.randn(2048, 2048, dtype=torch.float64, device="cuda:0")
st = torch.cuda.current_stream()
sh = st.cuda_stream
torch.cuda.synchronize()
det.stage_ingest(frames, B, grey, stream=sh); torch.mm(A, Bm); torch.cuda.synchronize()

def run(seq, reps=6):
    names = [n for n, _ in seq]
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(len(seq) + 1)] for _ in range(reps)]
    for r in range(reps):
        ev[r][0].record(st)
        for k, (_, fn) in enumerate(seq):
            fn(); ev[r][k + 1].record(st)
    torch.cuda.synchronize()
    for k, n in enumerate(names):
        print("   %-8s" % n, " ".join("%6.3f" % ev[r][k].elapsed_time(ev[r][k + 1]) for r in range(reps)))

ing = ("ingest", lambda: det.stage_ingest(frames, B, grey, stream=sh))
ing2 = ("ingest2", lambda: det.stage_ingest(frames, B, grey2, stream=sh))
den = ("dense", lambda: det.stage_threshold_corner(grey, B, binm, cand, cnt, stream=sh))
prb = ("probe", lambda: torch.mm(A, Bm))
for title, seq in [("probe alone", [prb]), ("dense alone", [den]), ("ingest alone", [ing]), ("ingest, probe", [ing, prb]), ("dense, probe", [den, prb]),
                   ("ingest, dense", [ing, den]), ("ingest (other buffer), dense", [ing2, den]), ("ingest, probe, dense", [ing, prb, den]),
                   ("ingest, dense, dense", [ing, den, ("dense", den[1])])]:
    print(title); run(seq)
