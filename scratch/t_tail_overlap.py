"""Experiment (experiments library): the lattice + pose kernel of a streamed batch on its own stream under the next batch's ingest pass
(rcc_set_tail_overlap).  First: the streamed records with the overlap on equal the synchronous call's, batch by batch; then frames/s
with it off / on, alternating.  usage: RCC_LIBRARY=.../librcc_hip_exp.so python scratch/t_tail_overlap.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robot_camera_calibration_amd import abi, api, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = api.default_config(); abi.set_geometry(cfg, 1920, 1080, abi.RCC_PIX_BGR8); cfg.batch_capacity = B
det = api.Detector(cfg)
sp = abi.default_synth_params()
sets = []
for k in range(2):
    poses = synth.sample_poses(B, cfg, first_index=k * B)
    fr = torch.empty((B, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
    for s0 in range(0, B, 64):
        det.synth_render(sp, poses[s0:s0 + 64], fr[s0:s0 + 64], first_index=k * B + s0)
    sets.append(fr)
torch.cuda.synchronize()
ref = [det.detect(f, B, want_corners=False)[0].copy() for f in sets]
det.set_tail_overlap(1)
order = [0, 1, 1, 0, 1, 0, 0, 1]
det.submit(sets[order[0]], B)
bad = 0
for i in range(len(order)):
    if i + 1 < len(order): det.submit(sets[order[i + 1]], B)
    d, _ = det.collect()
    r = ref[order[i]]
    same = len(d) == len(r) and d.tobytes() == r.tobytes()
    bad += not same
print("overlap on: %d of %d streamed batches differ from the synchronous call's records (%d records each)" % (bad, len(order), len(ref[0])))
def run(mode, K=20, W=5):
    det.set_tail_overlap(mode)
    f = sets[0]
    for phase, n in (("w", W), ("t", K)):
        if phase == "t":
            torch.cuda.synchronize(); t0 = time.perf_counter()
        det.submit(f, B)
        for k in range(n):
            if k + 1 < n: det.submit(f, B)
            det.collect()
    torch.cuda.synchronize()
    return B * K / (time.perf_counter() - t0)
for rep in range(3):
    a = run(0); b = run(1)
    print("frames/s: overlap off %.0f   on %.0f   (%+.1f %%)" % (a, b, 100 * (b / a - 1)))
det.close()
sys.exit(1 if bad else 0)
