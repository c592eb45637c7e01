#!/usr/bin/env python3
"""Helper of tests/test_experiments_library.py (not a test module): runs in a child process whose RCC_LIBRARY points at
librcc_hip_exp.so and checks the measurement-only forms of the threshold + corner pass that only that library carries --
the two-kernel variant 3 (band sweep + k_dense_runs on the active rows), the gang form of k_dense_wave, the 128 x 8 ingest
tiles of rounds 1-3 and the lattice + pose kernel overlapped with the next batch's ingest pass -- for bit-identity
with the forms the product library runs.  Prints one digest per geometry of the default form's outputs; the parent compares
them with the product library's on the same frames."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

GEOMS = (("bgr", 640, 480, 4), ("bgr", 1920, 1080, 3), ("mono", 3840, 2160, 2), ("mono", 2064, 1160, 2))


def digest_default(torch, abi, api, synth, kind, w, h, n, exp):
    from tests.util import sorted_cands
    cfg = api.default_config()
    abi.set_geometry(cfg, w, h, abi.RCC_PIX_BGR8 if kind == "bgr" else abi.RCC_PIX_MONO8)
    cfg.batch_capacity = n
    det = api.Detector(cfg)
    sp = abi.default_synth_params(seed=99)
    poses = synth.sample_poses(n, cfg, seed=99)
    frames = torch.empty((n, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
    det.synth_render(sp, poses, frames)
    px = w * h

    def stage(dv, skip, iv=-1):
        det.set_dense_variant(dv); det.set_dense_skip(skip); det.set_ingest_variant(iv)
        grey = torch.zeros((n, px), dtype=torch.uint8, device="cuda:0"); binm = torch.zeros((n, px), dtype=torch.uint8, device="cuda:0")
        cand = torch.zeros((n, cfg.max_candidates * 8), dtype=torch.uint8, device="cuda:0"); cnt = torch.zeros((n,), dtype=torch.int32, device="cuda:0")
        torch.cuda.synchronize()
        det.stage_ingest(frames, n, grey)
        det.stage_threshold_corner(grey, n, binm, cand, cnt)
        c = cand.cpu().numpy().view(api.CAND_DT).reshape(n, cfg.max_candidates); k = cnt.cpu().numpy()
        return binm.cpu().numpy(), [sorted_cands(c[f][:k[f]]) for f in range(n)], k, det.last_dense_kernel(), grey.cpu().numpy()
    ref = stage(-1, 1)
    det.set_dense_variant(-1); det.set_dense_skip(1)
    d0, f0 = det.detect(frames, n)
    img0 = det.fetch_images(n)
    hsh = hashlib.sha256()
    for a in (ref[0], ref[2], ref[4], d0, f0, img0["bin"], img0["cand_count"]):
        hsh.update(np.ascontiguousarray(a).tobytes())
    for c in ref[1]:
        hsh.update(np.ascontiguousarray(c).tobytes())
    if exp:
        o = stage(-1, 1, 3)                                 # the 128 x 8 ingest tiles of rounds 1-3 (the product runs 128 x 16)
        assert (o[4] == ref[4]).all() and (o[0] == ref[0]).all(), "ingest variant 3: grey / binary image differ at %dx%d" % (w, h)
        det.set_ingest_variant(-1)
        for dv, skip in ((3, 1), (3, 0)):                  # the two-kernel form, through the stage call and through detect()
            o = stage(dv, skip)
            assert "k_dense_runs" in o[3], o[3]
            assert (o[0] == ref[0]).all() and (o[2] == ref[2]).all(), "variant 3 (skip %d): binary image / counts differ at %dx%d" % (skip, w, h)
            for f in range(n):
                assert (o[1][f] == ref[1][f]).all(), "variant 3: candidates differ (frame %d)" % f
            det.set_dense_variant(dv); det.set_dense_skip(skip)
            d1, f1 = det.detect(frames, n)
            assert d1.tobytes() == d0.tobytes() and f1.tobytes() == f0.tobytes()
        det.set_dense_variant(-1); det.set_dense_skip(1)
        if w % 16 == 0 and w >= 256:
            for sync, seg in ((1, 0), (4, 3), (16, 2)):     # gangs of eight windows meeting every `sync` tile rows
                det.set_dense_gang(sync, seg)
                d1, f1 = det.detect(frames, n)
                assert "8>" in det.last_dense_kernel(), det.last_dense_kernel()
                img1 = det.fetch_images(n)
                assert d1.tobytes() == d0.tobytes() and f1.tobytes() == f0.tobytes()
                assert (img1["bin"] == img0["bin"]).all() and (img1["cand_count"] == img0["cand_count"]).all()
            det.set_dense_gang(0, 0)
        if kind == "bgr":
            # the lattice + pose kernel of a streamed batch on a stream of its own, under the next batch's ingest pass (rcc_set_tail_overlap):
            # five submissions, one batch ahead -- the records are the synchronous call's, byte for byte
            det.set_tail_overlap(1)
            det.submit(frames, n)
            for k in range(5):
                if k < 4:
                    det.submit(frames, n)
                d1, _ = det.collect()
                assert np.asarray(d1).tobytes() == np.asarray(d0).tobytes(), "tail overlap: streamed batch %d differs" % k
            det.set_tail_overlap(0)
    det.close()
    return hsh.hexdigest()


def main():
    import torch
    from robot_camera_calibration_amd import abi, api, synth
    exp = os.path.basename(api.library_path()) == "librcc_hip_exp.so"
    for kind, w, h, n in GEOMS:
        print("DIGEST %s %d %d %d %s" % (kind, w, h, n, digest_default(torch, abi, api, synth, kind, w, h, n, exp)), flush=True)


if __name__ == "__main__":
    main()
