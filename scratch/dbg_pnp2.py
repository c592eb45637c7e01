import sys, time
sys.path.insert(0, ".")
import numpy as np
from robot_camera_calibration_amd import abi, api, synth
from oracle import orc_py as O
cfg = api.default_config(); cfg.batch_capacity = 1
det = api.Detector(cfg)
K = np.array(list(cfg.K)); D = np.zeros(8)
obj = synth.board_object_points(8, 6, 0.108)
poses = synth.sample_poses(16, cfg, seed=31); rng = np.random.default_rng(1)
imgs = [synth.project_points(obj, p[:3], p[3:], K) + rng.normal(0, 0.05, (48, 2)) for p in poses]
ref = [O.solve_pnp(obj, im, K, 0, D) for im in imgs]
big_obj = [obj] * 1024; big_img = [imgs[i % 16] for i in range(1024)]
for v in (0, 1):
    det.set_pnp_variant(v)
    r, t, rms, st, it = det.solve_pnp([obj]*16, imgs, K, D, abi.RCC_DIST_NONE)
    worst = max(max(np.abs(r[k]-ref[k][1]).max(), np.abs(t[k]-ref[k][2]).max()) for k in range(16))
    det.solve_pnp(big_obj, big_img, K, D, abi.RCC_DIST_NONE)
    t0 = time.perf_counter(); det.solve_pnp(big_obj, big_img, K, D, abi.RCC_DIST_NONE); dt = time.perf_counter() - t0
    print("variant", v, "max diff vs oracle %.2e" % worst, "iters", list(it[:6]), " 1024x48pt solve incl. copies: %.2f ms" % (dt*1e3), flush=True)
