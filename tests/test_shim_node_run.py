"""N3, run for real (round 4): the ROS node's own translation unit (host/tag_detections_shim.cpp) compiled with g++ against the
stand-in headers of tests/host/mock_ros/ AND linked with librcc_hip.so into an executable, which a process-local message bus
(tests/host/mock_ros/mock_spin.cpp) feeds from a bag file -- camera_info, then image messages -- and whose publications the test reads
back.  Until round 4 the node had only been syntax-checked.

What it shows: the compiled host above the C ABI (parameters in ROS's remapping syntax, encoding policy, intrinsics from camera_info or
from the rosparams of camera_pose.cpp:59-64, handle rebuilt when the image geometry changes, rcc_detect_batch on host memory, message
filling, overlay image) yields, message for message, the records the Python host gets from the same library for the same frames.
What it does not show: anything about roscpp (transport, queues, timing) -- the bus is 150 lines of test code.

CPU part: the node's refusals (encoding, malformed image, no intrinsics) need no device, and without one rcc_create's error is
reported and the node keeps running."""
import os
import struct
import subprocess

import numpy as np
import pytest

from robot_camera_calibration_amd import abi, api, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "robot_camera_calibration_amd")
EXE = os.path.join(ROOT, "tests", "host", "rcc_detector_mock")
FAMILY = os.path.join(PKG, "data", "family36b.txt")


def build_node():
    srcs = [os.path.join(PKG, "host", "tag_detections_shim.cpp"), os.path.join(ROOT, "tests", "host", "mock_ros", "mock_spin.cpp")]
    deps = srcs + [os.path.join(PKG, "host", "tag_detections_fill.h"), os.path.join(ROOT, "include", "rcc.h"),
                   os.path.join(ROOT, "tests", "host", "mock_ros", "ros", "ros.h"), os.path.join(PKG, "librcc_hip.so")]
    if os.path.exists(EXE) and all(os.path.getmtime(EXE) >= os.path.getmtime(d) for d in deps):
        return EXE
    r = subprocess.run(["g++", "-std=c++14", "-O1", "-Wall", "-Wextra", "-Werror", *srcs, "-I", os.path.join(ROOT, "tests", "host", "mock_ros"),
                        "-I", os.path.join(ROOT, "include"), "-I", os.path.join(PKG, "host"), "-o", EXE,
                        "-L", PKG, "-lrcc_hip", "-Wl,-rpath," + PKG], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    return EXE


def write_bag(path, records):
    """records: ("I", K9, D) or ("F", seq, width, height, step, encoding, bytes)"""
    with open(path, "wb") as f:
        f.write(b"RCCBAG1\n")
        for r in records:
            if r[0] == "I":
                K, D = np.asarray(r[1], np.float64), np.asarray(r[2], np.float64)
                f.write(b"I" + K.tobytes() + struct.pack("<I", len(D)) + D.tobytes())
            else:
                _, seq, w, h, step, enc, data = r
                data = bytes(data)
                f.write(b"F" + struct.pack("<4I", seq, w, h, step) + struct.pack("<I", len(enc)) + enc.encode() + struct.pack("<Q", len(data)) + data)


def run_node(tmp_path, records, *args):
    bag, out = str(tmp_path / "in.bag"), str(tmp_path / "out.txt")
    write_bag(bag, records)
    if os.path.exists(out):
        os.remove(out)
    r = subprocess.run([build_node(), "__bag:=" + bag, "__out:=" + out, *args], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    msgs = []
    for line in open(out):
        t = line.split()
        if t[0] == "A":
            msgs.append(dict(kind="A", seq=int(t[1]), n=int(t[2]), dets=[]))
        elif t[0] == "D":
            msgs[-1]["dets"].append(dict(id=int(t[1]), size=float(t[2]), corners=np.array([float(v) for v in t[3:11]]).reshape(4, 2), t=np.array([float(v) for v in t[11:14]])))
        else:
            msgs.append(dict(kind=t[0], seq=int(t[1]), n=int(t[2]) if len(t) > 2 else 0))
    return msgs, r.stderr


K0 = [300.0, 0.0, 31.5, 0.0, 300.0, 23.5, 0.0, 0.0, 1.0]


def test_node_refusals_need_no_device(tmp_path):
    """the compiled node on whatever machine this is: an encoding it does not take and a malformed image are reported and skipped before
    anything touches the device; without intrinsics (neither rosparams nor camera_info) it says what camera_pose.cpp:67 says; and where
    rcc_create fails (no GPU here) the error is reported and the node lives on"""
    w, h = 64, 48
    grey = np.full((h, w, 3), 128, np.uint8).tobytes()
    recs = [("F", 0, w, h, 4 * w, "bgra8", bytes(4 * w * h)),           # four bytes per pixel: never read as BGR
            ("F", 1, w, h, 2 * w, "bgr8", grey),                          # step smaller than a row
            ("F", 2, w, h, 3 * w, "bgr8", grey[:100]),                    # fewer bytes than step x height
            ("F", 3, w, h, 3 * w, "bgr8", grey)]                          # well-formed, but no intrinsics anywhere
    msgs, err = run_node(tmp_path, recs, "_family_file:=" + FAMILY)
    assert [m["kind"] for m in msgs] == ["S"] * 4
    assert "encoding 'bgra8' is not supported" in err and err.count("malformed image") == 2
    assert "Camera intrinsics not loaded to parameter server!" in err
    # intrinsics on the parameter server in camera_pose_node's own form, and a family file that does not exist: no handle, a message
    msgs, err = run_node(tmp_path, recs[3:], "_family_file:=/nonexistent", "/camera_matrix/data:=" + ",".join(map(str, K0)), "/distortion_coefficients/data:=0,0,0,0,0")
    assert [m["kind"] for m in msgs] == ["S"] and "no tag family loaded" in err
    # everything in place: a device gives an (empty) array, no device gives rcc_create's error -- the process ends normally either way
    msgs, err = run_node(tmp_path, [("I", K0, [0.0] * 5)] + recs[3:], "_family_file:=" + FAMILY)
    assert len(msgs) >= 1 and ((msgs[0]["kind"] == "A" and msgs[0]["n"] == 0) or "rcc_create:" in err), (msgs, err)


@pytest.mark.gpu
def test_node_publishes_what_the_python_host_gets(tmp_path):
    """24 tags per 1280x720 frame through the compiled node: bgr8 twice (one handle), the same frame as rgb8, with padded rows (step >
    3 x width: the handle is rebuilt for the new stride), as mono8 (its green channel), an unsupported encoding in between; intrinsics
    once from camera_info and once from the rosparams.  Every published array equals the Python host's records for the same bytes:
    ids, sizes, the four corners bl, br, tr, tl and the pose's translation, bit for bit; the overlay image differs from its input on
    the outlines only."""
    import torch
    W, H, GX, GY = 1280, 720, 6, 4
    fam = abi.load_family()

    def make_cfg(pixfmt, stride=None):
        cfg = api.default_config()
        abi.set_geometry(cfg, W, H, pixfmt)
        if stride is not None:
            cfg.stride_bytes = stride
            cfg.frame_bytes = stride * H
        cfg.batch_capacity = 1
        abi.set_fiducial_target(cfg, fam, tag_size=0.10, max_hamming=2, max_targets=32, max_kept=abi.RCC_MAX_KEPT_FIDUCIAL)
        cfg.undistort = 0
        return cfg
    cfg = make_cfg(abi.RCC_PIX_BGR8)
    (hx, hy), centres, ids = synth.fiducial_grid_layout(GX, GY, cfg.tag_size)
    sp = abi.default_synth_params()
    sp.fid_grid_x, sp.fid_grid_y, sp.fid_gap_permille = GX, GY, 500
    det = api.Detector(cfg)
    poses = np.concatenate([synth.sample_poses(1, cfg, seed=900 + f, z_range=(0.9, 1.3), max_tilt_deg=30, half_extent_m=(hx, hy)) for f in range(2)])
    frames = torch.empty((2, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
    det.synth_render(sp, poses, frames)
    bgr = frames.cpu().numpy().reshape(2, H, W, 3)
    pad = 64
    padded = np.zeros((H, 3 * W + pad), np.uint8)
    padded[:, :3 * W] = bgr[0].reshape(H, 3 * W)
    variants = [("bgr8", 3 * W, bgr[0], make_cfg(abi.RCC_PIX_BGR8)), ("bgr8", 3 * W, bgr[1], None),
                ("rgb8", 3 * W, np.ascontiguousarray(bgr[0][..., ::-1]), make_cfg(abi.RCC_PIX_RGB8)),
                ("bgr8", 3 * W + pad, padded, make_cfg(abi.RCC_PIX_BGR8, 3 * W + pad)),
                ("mono8", W, np.ascontiguousarray(bgr[0][..., 1]), make_cfg(abi.RCC_PIX_MONO8))]
    expected = []
    d = None
    for enc, step, img, c in variants:
        if c is not None:
            if d is not None:
                d.close()
            d = api.Detector(c)
        dets, _ = d.detect(np.ascontiguousarray(img).reshape(1, -1), 1)
        expected.append(dets.copy())
    d.close(); det.close()
    assert len(expected[0]) == GX * GY and len(expected[2]) == GX * GY and len(expected[3]) == GX * GY and len(expected[4]) >= GX * GY - 2
    K, D = list(cfg.K), list(cfg.D)[:5]
    recs = [("I", K, D)]
    for seq, (enc, step, img, _) in enumerate(variants):
        recs.append(("F", seq, W, H, step, enc, np.ascontiguousarray(img).tobytes()))
        if seq == 1:
            recs.append(("F", 100, W, H, 2 * W, "yuv422", bytes(2 * W * H)))
    args = ["_family_file:=" + FAMILY, "_tag_size:=0.1", "_max_targets:=32", "_max_hamming:=2"]
    for how in ("camera_info", "rosparams"):
        extra = [] if how == "camera_info" else ["/camera_matrix/data:=" + ",".join(repr(float(v)) for v in K), "/distortion_coefficients/data:=" + ",".join(repr(float(v)) for v in D)]
        msgs, err = run_node(tmp_path, recs if how == "camera_info" else recs[1:], *args, *extra)
        assert ("intrinsics taken from camera_info" in err) == (how == "camera_info")
        assert "encoding 'yuv422' is not supported" in err
        arrays = [m for m in msgs if m["kind"] == "A"]
        overlays = [m for m in msgs if m["kind"] == "V"]
        assert [m["seq"] for m in arrays] == [0, 1, 2, 3, 4] and [m["seq"] for m in msgs if m["kind"] == "S"] == [100]
        for m, exp in zip(arrays, expected):
            assert m["n"] == len(exp) == len(m["dets"])
            for got, e in zip(m["dets"], exp):
                assert got["id"] == int(e.id) and got["size"] == float(e["size"])
                assert (got["corners"] == np.asarray(e.corners)).all() and (got["t"] == np.asarray(e.tvec)).all()
        # the overlay: some pixels changed, and no more than the outlines of the tags can hold (4 edges of <= 200 px each)
        assert [m["seq"] for m in overlays] == [0, 1, 2, 3, 4]
        for m, exp in zip(overlays, expected):
            assert 4 * 20 * len(exp) < m["n"] < 4 * 200 * len(exp)
