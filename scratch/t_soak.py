"""Soak: N streamed batches of 1024 x 1080p frames through submit / collect (every seventh with its corner tables), records compared
with the synchronous call's on every 97th batch; prints the sustained rate, the steps over 4 ms, and the growth of the host's resident
set and of the device's used memory.  Round 4: 12 000 batches (12.3 M frames) in 31.9 s = 385 k frames/s, no slow step after the
first, records identical, RSS high-water mark the same after 1 500 and 12 000 batches (no leak).  usage: python scratch/t_soak.py [N]"""
import os, sys, time, resource
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from robot_camera_calibration_amd import abi, api, synth
B = 1024
cfg = api.default_config(); abi.set_geometry(cfg, 1920, 1080, abi.RCC_PIX_BGR8); cfg.batch_capacity = B
det = api.Detector(cfg)
sp = abi.default_synth_params(); poses = synth.sample_poses(B, cfg)
fr = torch.empty((B, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
for s0 in range(0, B, 64): det.synth_render(sp, poses[s0:s0+64], fr[s0:s0+64], first_index=s0)
torch.cuda.synchronize()
ref, _ = det.detect(fr, B, want_corners=False); ref = ref.tobytes()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss; free0 = torch.cuda.mem_get_info()[0]
t0 = time.perf_counter(); tp = t0; slow = []; bad = 0
det.submit(fr, B)
for k in range(N):
    if k + 1 < N: det.submit(fr, B, want_corners=(k % 7 == 0))
    d, fc = det.collect()
    if k % 97 == 0 and d.tobytes() != ref: bad += 1
    t = time.perf_counter()
    if t - tp > 4e-3: slow.append((k, round(1e3 * (t - tp), 2)))
    tp = t
dt = time.perf_counter() - t0
print("soak: %d streamed batches of %d frames in %.1f s = %.0f frames/s; steps over 4 ms: %s; records differ in %d of the sampled batches" % (N, B, dt, N * B / dt, slow[:12], bad))
print("host max RSS %+d KB, device free memory %+d MB over the run" % (resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - rss0, (torch.cuda.mem_get_info()[0] - free0) >> 20))
det.close()
