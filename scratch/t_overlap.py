# experiment: ingest and threshold+corner pass launched together on two streams (independent buffers) vs back to back
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robot_camera_calibration_amd import abi, api, synth
B = 1024
cfg = api.default_config(); abi.set_geometry(cfg, 1920, 1080); cfg.batch_capacity = B
det = api.Detector(cfg)
frames = torch.empty((B, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
sp = abi.default_synth_params(); poses = synth.sample_poses(32, cfg)
poses = np.concatenate([poses] * (B // 32))
for s0 in range(0, B, 64):
    det.synth_render(sp, poses[s0:s0 + 64], frames[s0:s0 + 64], first_index=s0)
px = 1920 * 1080
g1 = torch.empty((B, px), dtype=torch.uint8, device="cuda:0"); g2 = torch.empty_like(g1)
cand = torch.empty((B, cfg.max_candidates * 8), dtype=torch.uint8, device="cuda:0"); cnt = torch.empty((B,), dtype=torch.int32, device="cuda:0")
torch.cuda.synchronize()
det.stage_ingest(frames, B, g2)
L = det._L
L.rcc_debug_overlap.argtypes = [C.c_void_p] * 2 + [C.c_int32] + [C.c_void_p] * 4 + [C.c_int32, C.c_int32, C.POINTER(C.c_float)]
for mode in (0, 1, 0, 1):
    ms = C.c_float(0)
    st = L.rcc_debug_overlap(det._h, api._ptr(frames), B, api._ptr(g1), api._ptr(g2), api._ptr(cand), api._ptr(cnt), mode, 5, C.byref(ms))
    print("mode", mode, "status", st, "ms per (ingest + dense)", round(ms.value, 3))
