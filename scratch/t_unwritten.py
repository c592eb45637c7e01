import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robot_camera_calibration_amd import abi, api, synth
for (W, H, B) in [(1920, 1080, 24), (3840, 2160, 8), (2064, 1080, 8), (640, 480, 8)]:
    cfg = api.default_config(); abi.set_geometry(cfg, W, H); cfg.batch_capacity = B
    det = api.Detector(cfg)
    frames = torch.empty((B, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
    sp = abi.default_synth_params(); poses = synth.sample_poses(B, cfg)
    det.synth_render(sp, poses, frames, first_index=0)
    px = W * H
    grey = torch.empty((B, px), dtype=torch.uint8, device="cuda:0")
    det.stage_ingest(frames, B, grey); torch.cuda.synchronize()
    for v in (0, 1, 2):
        det.set_dense_variant(v)
        b = torch.full_like(grey, 0x55); c = torch.empty((B, cfg.max_candidates * 8), dtype=torch.uint8, device="cuda:0"); n = torch.empty((B,), dtype=torch.int32, device="cuda:0")
        det.stage_threshold_corner(grey, B, b, c, n); torch.cuda.synchronize()
        bn = b.cpu().numpy().reshape(B, H, W)
        u = bn == 0x55
        msg = "%dx%d variant %d %s: unwritten pixels %d" % (W, H, v, det.last_dense_kernel(), int(u.sum()))
        if u.any():
            f, ys, xs = np.nonzero(u)
            msg += " frames %s y %d..%d x %d..%d" % (sorted(set(f.tolist()))[:6], ys.min(), ys.max(), xs.min(), xs.max())
        print(msg)
    det.close()
