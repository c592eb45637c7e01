import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robot_camera_calibration_amd import abi, api, synth
W, H, B = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = api.default_config(); abi.set_geometry(cfg, W, H); cfg.batch_capacity = B
det = api.Detector(cfg)
frames = torch.empty((B, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
sp = abi.default_synth_params(); poses = synth.sample_poses(B, cfg)
[det.synth_render(sp, poses[s0:s0 + 16], frames[s0:s0 + 16], first_index=s0) for s0 in range(0, B, 16)]
px = W * H
grey = torch.empty((B, px), dtype=torch.uint8, device="cuda:0")
det.stage_ingest(frames, B, grey); torch.cuda.synchronize()
g = torch.Generator(device="cuda:0"); g.manual_seed(7)
grey[: B // 2] = (grey[: B // 2].to(torch.int16) + torch.randint(-12, 13, (B // 2, px), device="cuda:0", generator=g, dtype=torch.int16)).clamp(0, 255).to(torch.uint8)
outs = {}
for v in (0, 1, 2):
    det.set_dense_variant(v)
    b = torch.zeros_like(grey); c = torch.empty((B, cfg.max_candidates * 8), dtype=torch.uint8, device="cuda:0"); n = torch.empty((B,), dtype=torch.int32, device="cuda:0")
    det.stage_threshold_corner(grey, B, b, c, n); torch.cuda.synchronize()
    outs[v] = (b.cpu().numpy().reshape(B, H, W), n.cpu().numpy(), det.last_dense_kernel())
    print(v, outs[v][2], "counts", outs[v][1][:12], "cap", cfg.max_candidates)
gn = grey.cpu().numpy().reshape(B, H, W)
for v in (1, 2):
    d = outs[v][0] != outs[0][0]
    print("variant", v, "differing pixels per frame", d.reshape(B, -1).sum(1))
    if d.any():
        f, y, x = np.argwhere(d)[0]
        print(" first diff frame %d y %d x %d: ref %d got %d grey %d; tile rows around:" % (f, y, x, outs[0][0][f, y, x], outs[v][0][f, y, x], gn[f, y, x]))
        ys, xs = np.nonzero(d[f]); print("  y range", ys.min(), ys.max(), "x range", xs.min(), xs.max(), "distinct (ref,got):", set(zip(outs[0][0][f][d[f]].tolist(), outs[v][0][f][d[f]].tolist())))
