#!/usr/bin/env python3
"""bench.py -- calibration frames/s at 1920x1080 on N MI355X (BASELINE.json metric).

A step = one pass of the whole hot path (ingest -> threshold+corner -> list/sub-pixel/board
indexing -> PnP -> result D2H) over one batch of 1024 synthetic 1920x1080 BGR8 checkerboard frames
that are already resident in HBM (BASELINE.json configs[1]).  With N ranks every rank owns its own
1024-frame batch (frames are independent units: weak scaling, no data-path collective) and the
per-frame pose records are all-gathered over RCCL once per step.

Prints ONE JSON line on rank 0; see DESIGN.md section 6 for the fields.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1024, help="frames per rank per step")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--cpu-sample", type=int, default=32, help="frames timed on the CPU oracle (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dense-variant", type=int, default=-1)
    ap.add_argument("--ingest-variant", type=int, default=-1)
    ap.add_argument("--roofline-reps", type=int, default=5)
    ap.add_argument("--sync-steps", action="store_true", help="time K synchronous detect() calls instead of the submit/collect stream of K batches")
    ap.add_argument("--pipeline", type=int, default=1, help="chunks of the detector's two-stream pipeline per step (1: single pass)")
    ap.add_argument("--fiducials", default="", help="BASELINE.json configs[4]-style run: GXxGY planar grid of square fiducials per frame, e.g. 6x4")
    ap.add_argument("--fisheye", action="store_true", help="BASELINE.json configs[3]-style run: fisheye model (use with --width 3840 --height 2160 --batch 256)")
    return ap.parse_args()


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=dev)   # "nccl" is RCCL on ROCm

    from robot_camera_calibration_amd import abi, api, synth
    from robot_camera_calibration_amd import dist as rdist

    cfg = api.default_config()
    abi.set_geometry(cfg, a.width, a.height, abi.RCC_PIX_BGR8)
    if a.fisheye:
        abi.set_distortion(cfg, abi.RCC_DIST_FISHEYE, abi.FISHEYE_DEFAULT)
    fid = None
    if a.fiducials:
        fid = tuple(int(v) for v in a.fiducials.lower().split("x"))
        family = abi.load_family()
        abi.set_fiducial_target(cfg, family, tag_size=0.10, max_targets=fid[0] * fid[1])
    cfg.device = local
    cfg.batch_capacity = a.batch
    det = api.Detector(cfg)
    det.set_pipeline(a.pipeline)
    det.set_dense_variant(a.dense_variant)
    det.set_ingest_variant(a.ingest_variant)
    B = a.batch
    px = a.width * a.height

    # ---- workload: B distinct frames per rank, rendered on the device (not timed)
    sp = abi.default_synth_params()
    first = rank * B
    frames = torch.empty((B, cfg.frame_bytes), dtype=torch.uint8, device=dev)
    if fid:
        (fhx, fhy), _, _ = synth.fiducial_grid_layout(fid[0], fid[1], cfg.tag_size)
        sp.fid_grid_x, sp.fid_grid_y, sp.fid_gap_permille = fid[0], fid[1], 500
        poses = synth.sample_poses(B, cfg, first_index=first, z_range=(1.0, 2.0), max_tilt_deg=40, half_extent_m=(fhx, fhy))
    else:
        poses = synth.sample_poses(B, cfg, first_index=first)

    chunk = 64
    for s0 in range(0, B, chunk):
        n = min(chunk, B - s0)
        det.synth_render(sp, poses[s0:s0 + n], frames[s0:s0 + n], first_index=first + s0)
    torch.cuda.synchronize()

    gather = rdist.PoseGather(B * (fid[0] * fid[1] if fid else 1), dev, world, dist)

    def step():
        dets, _ = det.detect(frames, B, want_corners=False)
        return gather.run(dets)

    for _ in range(a.warmup):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    found = 0
    if a.sync_steps:
        for _ in range(a.steps):
            found = step()
    else:
        # the streaming form of the same K steps (rcc_detect_batch_submit / _collect): batch k+1 is launched before
        # the host unpacks batch k, so the device does not idle during the unpack and the exchange of the records.
        # Every step's work -- all kernels, the device-to-host copy, the unpack, the all_gather -- is inside the region.
        det.submit(frames, B)
        for k in range(a.steps):
            if k + 1 < a.steps:
                det.submit(frames, B)
            dets, _ = det.collect()
            found = gather.run(dets)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    # the same K steps as synchronous detect() calls, for comparison (reported, not `value`)
    sync_fps = None
    if not a.sync_steps:
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t1s = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dts = torch.tensor([time.perf_counter() - t1s], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(dts, op=dist.ReduceOp.MAX)
        sync_fps = world * B * a.steps / float(dts.item())
    # per-stage times: one extra (untimed) step as a single pass on one stream -- in the pipelined step the stages of
    # different chunks overlap, so they have no separate durations
    prev = det.set_pipeline(1)
    step()
    timings = det.last_timings()
    det.set_pipeline(prev)

    out = None
    if rank == 0:
        fps = world * B * a.steps / dt
        out = {
            "metric": "calibration frames/sec at %dx%d" % (a.width, a.height), "value": fps, "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/i32 pixel stages, f64 sub-pixel + PnP", "data": "synthetic",
            "config": {"workload": "batch of %d synthetic %dx%d BGR8 checkerboard frames per GPU, device-resident "
                                   "(BASELINE.json configs[1]); corners + PnP" % (B, a.width, a.height),
                       "frames_per_step_per_gpu": B,
                       "target": ("%dx%d square fiducials of 0.10 m per frame (build family36b), 4-point PnP per tag" % fid) if fid else "8x6 inner-corner checkerboard, 0.108 m", "distortion": ("fisheye" if a.fisheye else "plumb-bob") + ", undistort on",
                       "parallelism": "frame-sharded, 1 process per GPU, 1 all_gather of pose records per step"},
            "targets_found_in_last_step": int(found), "targets_expected_per_step": int(world * B * (fid[0] * fid[1] if fid else 1)), "stage_ms_single_pass": timings, "pipeline_chunks": a.pipeline, "step_form": "sync detect()" if a.sync_steps else "submit/collect, one batch ahead", "value_with_sync_steps": sync_fps,
        }

    # ---- roofline of the threshold+corner pass (the kernel BASELINE.json's north_star names) and
    # of the ingest pass: algorithmic bytes / HIP-event time on the launch stream
    if rank == 0:
        grey = torch.empty((B, px), dtype=torch.uint8, device=dev)
        binm = torch.empty((B, px), dtype=torch.uint8, device=dev)
        cand = torch.empty((B, cfg.max_candidates * 8), dtype=torch.uint8, device=dev)
        cnt = torch.empty((B,), dtype=torch.int32, device=dev)
        det.stage_ingest(frames, B, grey)
        det.time_dense(grey, B, binm, cand, cnt, 1)
        ms = det.time_dense(grey, B, binm, cand, cnt, a.roofline_reps)
        alg = 2.0 * px * B
        ach = alg / (ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_dense.json")
        if os.path.exists(tpath) and (a.width, a.height) == (1920, 1080):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_frame") * B   # PMC-derived, per frame x frames per launch
            except Exception:
                traffic = None
        # the form the detect path runs: binary image left as a per-tile threshold map (2*px -> (1 + 1/16)*px bytes)
        det.time_dense(grey, B, None, cand, cnt, 1)
        ms_c = det.time_dense(grey, B, None, cand, cnt, a.roofline_reps)
        # yardstick measured in the same process: a plain streaming copy of the same bytes (grey -> binary buffer)
        copy_ms = det.time_copy(grey, binm, B * px, a.roofline_reps) if (B * px) % 16 == 0 else None
        out["roofline"] = {"bound": "hbm", "kernel": "threshold+corner pass (k_dense_*)", "achieved": ach, "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "frac_of_guide_copy_6290": ach / 6290.0,
                           "traffic": traffic, "alg_bytes_per_launch": alg, "ms_per_launch": ms, "frames_per_launch": B,
                           "copy_same_bytes_ms": copy_ms, "copy_GBps": (alg / (copy_ms * 1e-3) / 1e9) if copy_ms else None,
                           "frac_of_copy": (copy_ms / ms) if copy_ms else None,
                           "detect_path_variant": {"what": "same pass, binary image kept as a 1-byte-per-4x4-tile threshold map (what rcc_detect_batch runs)",
                                                   "ms_per_launch": ms_c, "alg_bytes_per_launch": (1.0 + 1.0 / 16.0) * px * B,
                                                   "achieved": (1.0 + 1.0 / 16.0) * px * B / (ms_c * 1e-3) / 1e9}}
        det.time_ingest(frames, B, grey, 1)
        msi = det.time_ingest(frames, B, grey, max(1, a.roofline_reps // 2))
        algi = 4.0 * px * B
        traffic_i = None
        tpath = os.path.join(ROOT, "profiles", "traffic_ingest.json")
        if os.path.exists(tpath) and (a.width, a.height) == (1920, 1080) and not a.fisheye:
            try:
                traffic_i = json.load(open(tpath)).get("hbm_bytes_per_frame") * B
            except Exception:
                traffic_i = None
        out["roofline_ingest"] = {"bound": "hbm", "kernel": "undistort+grey (k_ingest_*)", "achieved": algi / (msi * 1e-3) / 1e9,
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": algi / (msi * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "traffic": traffic_i, "alg_bytes_per_launch": algi, "ms_per_launch": msi}
        del grey, binm, cand, cnt

    # ---- CPU baseline (the oracle = "port"; the reference's OpenCV path cannot be built here) and
    # accuracy against it, on a bounded sample, rank 0 at N=1 only
    if rank == 0 and world == 1 and not a.no_cpu_baseline and not fid:
        from concurrent.futures import ThreadPoolExecutor
        from oracle import orc_py
        S = min(a.cpu_sample, B)
        host = frames[:S].cpu().numpy()
        dets, fcs = det.detect(frames[:S].contiguous(), S, want_corners=True)
        by = {int(d.frame): d for d in dets}
        T = max(1, min(os.cpu_count() or 1, 16, S))
        ctxs = [orc_py.Context(cfg) for _ in range(T)]
        parts = [list(range(t, S, T)) for t in range(T)]
        res = [None] * S

        def work(t):
            for f in parts[t]:
                res[f] = ctxs[t].detect(host[f], f)
        t1 = time.perf_counter()
        with ThreadPoolExecutor(T) as ex:
            list(ex.map(work, range(T)))
        cdt = time.perf_counter() - t1
        # single-thread rate on a few of the same frames (SURVEY 8(d) asks for both)
        S1 = min(8, S)
        t2 = time.perf_counter()
        for f in range(S1):
            ctxs[0].detect(host[f], f)
        cdt1 = time.perf_counter() - t2
        mxc = mxr = mxt = 0.0
        gtc = gtr = gtt = 0.0
        Kb = np.array(list(cfg.K)); objb = synth.board_object_points(cfg.board_cols, cfg.board_rows, cfg.board_square)
        mism = 0
        nc = cfg.board_cols * cfg.board_rows
        for f in range(S):
            n, od, ofc = res[f]
            if (ofc.ncorners != fcs[f].ncorners) or (ofc.status != fcs[f].status):
                mism += 1
                continue
            if n:
                gp = np.array([[fcs[f].px[k][0], fcs[f].px[k][1]] for k in range(nc)])
                op = np.array([[ofc.px[k][0], ofc.px[k][1]] for k in range(nc)])
                mism += int((gp != op).any())
                gx = np.array([[fcs[f].xy[k][0], fcs[f].xy[k][1]] for k in range(nc)])
                ox = np.array([[ofc.xy[k][0], ofc.xy[k][1]] for k in range(nc)])
                mxc = max(mxc, float(np.abs(gx - ox).max()))
                mxr = max(mxr, float(np.abs(np.array(list(by[f].rvec)) - np.array(list(od.rvec))).max()))
                mxt = max(mxt, float(np.abs(np.array(list(by[f].tvec)) - np.array(list(od.tvec))).max()))
                # informational: against the analytic ground truth of the synthetic camera (undistorted image =
                # pinhole projection of the board; the 9x7-square board has a 180-degree ambiguity)
                gt = synth.project_points(objb, poses[f][:3], poses[f][3:], Kb)
                flip = np.abs(gx - gt).max() > np.abs(gx - gt[::-1]).max()
                gtc = max(gtc, float(np.abs(gx - (gt[::-1] if flip else gt)).max()))
                Rg = synth.rodrigues(poses[f][:3]) @ (np.diag([-1.0, -1.0, 1.0]) if flip else np.eye(3))
                gtr = max(gtr, float(np.abs(synth.rodrigues(list(by[f].rvec)) - Rg).max()))
                gtt = max(gtt, float(np.abs(np.array(list(by[f].tvec)) - poses[f][3:]).max()))
        out["cpu_baseline"] = {"value": S / cdt, "unit": "frames/s", "cores": T, "single_thread_value": S1 / cdt1, "kind": "port",
                               "sample": "%d of the same 1920x1080 frames through oracle/ (C, -O2, %d threads over frames); "
                                         "host has %d logical CPUs" % (S, T, os.cpu_count() or 0)}
        out["accuracy_vs_oracle"] = {"frames": S, "max_corner_err_px": mxc, "max_rvec_err": mxr, "max_tvec_err": mxt,
                                     "corner_index_or_status_mismatches": mism}
        out["accuracy_vs_ground_truth"] = {"frames": S, "max_corner_err_px": gtc, "max_rotation_matrix_err": gtr, "max_tvec_err_m": gtt,
                                           "note": "informational: detector error on noisy supersampled renders, not a parity figure"}

    if rank == 0:
        print(json.dumps(out))
    det.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
