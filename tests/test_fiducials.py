"""Square-fiducial form of a4/a6 + per-tag a7 (BASELINE.json configs[4]: >= 16 fiducials per frame).

CPU: the oracle against analytic ground truth on rendered multi-tag scenes (ids, corner order
bl,br,tr,tl as real_preprocessing/src/camera_pose.cpp:123-126,152-155, per-tag pose).
GPU: the HIP path against the oracle: integer stages bit-exact, ids / corner order identical,
corners and poses within 1e-4."""
import numpy as np
import pytest

from robot_camera_calibration_amd import abi, api, synth

GX, GY = 6, 4


def _cfg(factory, w=1280, h=720, B=3, refine=abi.RCC_TAG_REFINE_EDGES):
    cfg = factory()
    abi.set_geometry(cfg, w, h, abi.RCC_PIX_BGR8)
    cfg.batch_capacity = B
    fam = abi.load_family()
    abi.set_fiducial_target(cfg, fam, tag_size=0.10)
    cfg.tag_refine = refine
    return cfg, fam


REFINE_MODES = pytest.mark.parametrize("refine", [abi.RCC_TAG_REFINE_EDGES, abi.RCC_TAG_REFINE_CORNER_SUBPIX], ids=["refine_edges", "corner_subpix"])


def _scene(cfg):
    (hx, hy), centres, ids = synth.fiducial_grid_layout(GX, GY, cfg.tag_size)
    sp = abi.default_synth_params()
    sp.fid_grid_x, sp.fid_grid_y, sp.fid_gap_permille = GX, GY, 500
    return (hx, hy), centres, ids, sp


def test_family_properties():
    fam = abi.load_family()
    assert len(fam) == 48 and len(set(int(c) for c in fam)) == 48

    def rot(c):
        o = 0
        for r in range(6):
            for cc in range(6):
                o |= ((c >> (35 - (cc * 6 + (5 - r)))) & 1) << (35 - (r * 6 + cc))
        return o
    best = 36
    for i, a in enumerate(int(c) for c in fam):
        rs = [a]
        for _ in range(3):
            rs.append(rot(rs[-1]))
        assert rot(rs[-1]) == a
        best = min(best, min(bin(rs[0] ^ r).count("1") for r in rs[1:]))
        for b in (int(c) for c in fam[:i]):
            best = min(best, min(bin(r ^ b).count("1") for r in rs))
    assert best >= 10        # <= 2 bit errors can never reach another code or rotation


@REFINE_MODES
def test_oracle_fiducials_against_ground_truth(oracle, refine):
    """ids, corner order and pose against the renderer's ground truth, for both corner refinements.  The refine_edges form
    (the default) puts the corners within 0.3 px (rms 0.04) and the tag position within 4 mm; the
    cornerSubPix form, run at an L-corner, within 0.6 px / 2 cm -- the corners go straight into a 4-point pose
    (camera_pose.cpp:152-163), so the corner error is the pose error."""
    cfg, fam = _cfg(oracle.default_config, refine=refine)
    edges = refine == abi.RCC_TAG_REFINE_EDGES
    (hx, hy), centres, ids, sp = _scene(cfg)
    K = np.array(list(cfg.K))
    ctx = oracle.Context(cfg)
    objt = synth.tag_object_points(cfg.tag_size)
    for f in range(3):
        pose = synth.sample_poses(1, cfg, seed=100 + f, z_range=(0.9, 1.6), max_tilt_deg=40, half_extent_m=(hx, hy))[0]
        img = oracle.synth_render(cfg, sp, pose, f)
        n, det, fc = ctx.detect(img, f)
        assert n == GX * GY
        got = {det[k].id: det[k] for k in range(n)}
        assert sorted(got) == list(ids)
        R = synth.rodrigues(pose[:3])
        for c, i in zip(centres, ids):
            d = got[i]
            assert d.ncorners == 4 and d.hamming == 0 and d.size == cfg.tag_size and d.pnp_status == 0
            gt = synth.project_points(objt + c, pose[:3], pose[3:], K)          # bl, br, tr, tl
            assert np.abs(np.array([[d.corners[q][0], d.corners[q][1]] for q in range(4)]) - gt).max() < (0.3 if edges else 0.6)
            assert np.abs(np.array(d.tvec[:]) - (R @ c + pose[3:])).max() < (0.004 if edges else 0.02)
            assert np.abs(synth.rodrigues(list(d.rvec)) - R).max() < (0.03 if edges else 0.08)


@pytest.mark.gpu
@pytest.mark.parametrize("optics", [None, (1.0, 300, -200, 400), (1.5, 0, 0, 0), ("3tap", -300, 200, 300)], ids=["ideal", "g1.0_shaded", "g1.5", "3tap_shaded"])
@REFINE_MODES
def test_hip_fiducials_match_oracle(oracle, refine, optics):
    """optics: the same through the synthetic camera's blur / illumination gradient / vignette (rcc_synth_params, ABI 2)"""
    import torch
    cfg, fam = _cfg(api.default_config, B=3, refine=refine)
    (hx, hy), centres, ids, sp = _scene(cfg)
    if optics:
        abi.set_optics(sp, *optics)
    det = api.Detector(cfg)
    n = 3
    poses = np.concatenate([synth.sample_poses(1, cfg, seed=200 + f, z_range=(0.9, 1.6), max_tilt_deg=40, half_extent_m=(hx, hy)) for f in range(n)])
    frames = torch.empty((n, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
    det.synth_render(sp, poses, frames)
    dets, fcs = det.detect(frames, n)
    img = det.fetch_images(n)
    lst = det.fetch_lists(n)
    host = frames.cpu().numpy()
    ctx = oracle.Context(cfg)
    k0 = 0
    worst = 0.0
    for f in range(n):
        m, odet, ofc, st = ctx.detect(host[f], f, stages=True)
        assert (img["grey"][f] == st["grey"]).all() and (img["bin"][f] == st["bin"]).all()
        assert lst["npre"][f] == st["npre"]
        p = lst["pre"][f][:st["npre"]]
        assert (p["x"] == st["pre"]["x"]).all() and (p["y"] == st["pre"]["y"]).all()
        assert np.abs(lst["pre_xy"][f][:st["npre"]] - st["pre_xy"]).max() == 0.0
        mine = dets[k0:k0 + m]
        assert (m == GX * GY or optics) and m >= GX * GY - 2 and len(mine) == m and (mine.frame == f).all()
        for k in range(m):
            a, b = mine[k], odet[k]
            assert a.id == b.id and a.hamming == b.hamming and a.ncorners == 4 and a.pnp_status == b.pnp_status
            assert np.abs(a.corners - np.array([[b.corners[q][0], b.corners[q][1]] for q in range(4)])).max() == 0.0
            worst = max(worst, np.abs(a.rvec - np.array(b.rvec[:])).max(), np.abs(a.tvec - np.array(b.tvec[:])).max())
        k0 += m
    assert k0 == len(dets) and worst <= 1e-4
    print("fiducials: %d tags, max pose diff vs oracle %.2e" % (k0, worst))
    det.close()



@pytest.mark.gpu
def test_hip_subpix_grid_width_does_not_change_results():
    """tag scenes: the sub-pixel kernel's grid is narrower than the candidate list and a wave walks the list (the automatic width
    only gets below the list's length on batches of hundreds of frames: rcc_set_subpix_grid forces it here).  Widths 1, 7, 64 and
    the full list give the same refined positions and the same detections, bit for bit -- also for the candidates that leave at
    the convex-black-corner test and keep their pixel."""
    import torch
    cfg, fam = _cfg(api.default_config, B=3)
    (hx, hy), centres, ids, sp = _scene(cfg)
    det = api.Detector(cfg)
    n = 3
    poses = np.concatenate([synth.sample_poses(1, cfg, seed=200 + f, z_range=(0.9, 1.6), max_tilt_deg=40, half_extent_m=(hx, hy)) for f in range(n)])
    frames = torch.empty((n, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
    det.synth_render(sp, poses, frames)
    ref = None
    for width in (0, 1, 7, 64, 100000):
        det.set_subpix_grid(width)
        dets, fcs = det.detect(frames, n)
        lst = det.fetch_lists(n)
        assert (lst["npre"] > 300).all()                                    # far more candidates than the narrow grids are wide
        got = (dets.tobytes(), b"".join(lst["pre_xy"][f][:lst["npre"][f]].tobytes() for f in range(n)), lst["npre"].tobytes())
        if ref is None:
            ref = got
            assert len(dets) == n * GX * GY
            kept_pixel = sum(int((lst["pre_xy"][f][:lst["npre"][f]] == np.stack([lst["pre"][f]["x"], lst["pre"][f]["y"]], 1)[:lst["npre"][f]]).all(axis=1).sum()) for f in range(n))
            assert kept_pixel > int(lst["npre"].sum()) // 3                 # the corner test in front of the refinement sends many home
        assert got == ref, "grid width %d" % width
    det.close()

def _clutter(img, w, h, seed, count):
    """paste black / white / grey rectangles over a BGR frame (destroys some tags: only equality with the oracle is asked)"""
    rng = np.random.default_rng(seed)
    a = img.reshape(h, w, 3).copy()
    for _ in range(count):
        rw, rh = int(rng.integers(8, 48)), int(rng.integers(8, 48))
        x, y = int(rng.integers(0, w - rw)), int(rng.integers(0, h - rh))
        a[y:y + rh, x:x + rw, :] = int(rng.choice([10, 60, 200, 245]))
    return a.reshape(-1)


@pytest.mark.gpu
@pytest.mark.parametrize("clutter", [0, 40, 120, 400])
def test_hip_fiducials_dense_and_cluttered_scenes(oracle, clutter):
    """48 tags at 1920x1080 (more than 256 classified corners: several passes of the linking loop) and the same scenes
    under hundreds of pasted rectangles (corner list at its cap, most classified corners link to nothing: the pair-queue
    fallback, pruning by the running best, quads that fail to decode): detections identical to the oracle's."""
    import torch
    gx, gy = 8, 6
    cfg, fam = _cfg(api.default_config, w=1920, h=1080, B=2)
    (hx, hy), centres, ids = synth.fiducial_grid_layout(gx, gy, cfg.tag_size)
    sp = abi.default_synth_params()
    sp.fid_grid_x, sp.fid_grid_y, sp.fid_gap_permille = gx, gy, 500
    det = api.Detector(cfg)
    n = 2
    poses = np.concatenate([synth.sample_poses(1, cfg, seed=300 + f, z_range=(1.5, 1.9), max_tilt_deg=30, half_extent_m=(hx, hy)) for f in range(n)])
    frames = torch.empty((n, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
    det.synth_render(sp, poses, frames)
    host = frames.cpu().numpy()
    if clutter:
        host = np.stack([_clutter(host[f], 1920, 1080, 7 * clutter + f, clutter) for f in range(n)])
        frames.copy_(torch.from_numpy(host).to("cuda:0"))
        torch.cuda.synchronize()
    dets, fcs = det.detect(frames, n)
    lst = det.fetch_lists(n)
    ctx = oracle.Context(cfg)
    k0, total, classified_max, overflowed = 0, 0, 0, 0
    for f in range(n):
        m, odet, ofc, st = ctx.detect(host[f], f, stages=True)
        assert int(fcs[f].status) == int(ofc.status)
        overflow = int(ofc.status) & (abi.RCC_FRAME_CAND_OVERFLOW | abi.RCC_FRAME_KEPT_OVERFLOW)
        if overflow:                          # more corners than max_kept: the frame yields nothing, on both sides
            assert m == 0
            overflowed += 1
        else:
            assert lst["npre"][f] == st["npre"]
            assert np.abs(lst["pre_xy"][f][:st["npre"]] - st["pre_xy"]).max() == 0.0
        mine = dets[k0:k0 + m]
        assert len(mine) == m and (mine.frame == f).all(), (len(mine), m)
        for k in range(m):
            a, b = mine[k], odet[k]
            assert a.id == b.id and a.hamming == b.hamming and a.ncorners == 4 and a.pnp_status == b.pnp_status
            assert np.abs(a.corners - np.array([[b.corners[q][0], b.corners[q][1]] for q in range(4)])).max() == 0.0
            assert np.abs(a.rvec - np.array(b.rvec[:])).max() <= 1e-4 and np.abs(a.tvec - np.array(b.tvec[:])).max() <= 1e-4
        k0 += m
        total += m
        classified_max = max(classified_max, int(st["npre"]))
    assert k0 == len(dets)
    if clutter == 0:
        assert total == n * gx * gy
    print("clutter %d: %d tags, up to %d refined corners per frame, %d frames over the cap" % (clutter, total, classified_max, overflowed))
    det.close()


@pytest.mark.gpu
def test_hip_fiducials_random_pose_sweep(oracle):
    """48 random poses (tilt up to 55 degrees, some tags leave the frame or get too small to decode): per frame the same
    detections as the oracle, in the same order."""
    import torch
    n = 48
    cfg, fam = _cfg(api.default_config, B=n)
    (hx, hy), centres, ids, sp = _scene(cfg)
    det = api.Detector(cfg)
    poses = np.concatenate([synth.sample_poses(1, cfg, seed=900 + f, z_range=(0.8, 2.6), max_tilt_deg=55, half_extent_m=(hx * 0.7, hy * 0.7))
                            for f in range(n)])
    frames = torch.empty((n, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
    det.synth_render(sp, poses, frames)
    dets, fcs = det.detect(frames, n)
    host = frames.cpu().numpy()
    ctx = oracle.Context(cfg)
    k0, worst, partial = 0, 0.0, 0
    for f in range(n):
        m, odet, ofc = ctx.detect(host[f], f)
        assert int(fcs[f].status) == int(ofc.status)
        mine = dets[k0:k0 + m]
        assert len(mine) == m and (mine.frame == f).all(), (f, len(mine), m)
        for k in range(m):
            a, b = mine[k], odet[k]
            assert a.id == b.id and a.hamming == b.hamming and a.pnp_status == b.pnp_status
            assert np.abs(a.corners - np.array([[b.corners[q][0], b.corners[q][1]] for q in range(4)])).max() == 0.0
            worst = max(worst, np.abs(a.rvec - np.array(b.rvec[:])).max(), np.abs(a.tvec - np.array(b.tvec[:])).max())
        k0 += m
        partial += m != GX * GY
    assert k0 == len(dets) and worst <= 1e-4
    print("sweep: %d tags over %d frames (%d frames with fewer than %d), max pose diff %.2e" % (k0, n, partial, GX * GY, worst))
    det.close()


@pytest.mark.gpu
def test_record_table_multi_tag_frames():
    """the device-packed record table (rcc_set_record_tables) with several tags per frame: slot = frame * max_targets + q,
    equal to the host form dist.pack builds from the same records; a multi-tag frame survives the exchange whole"""
    import torch
    from robot_camera_calibration_amd import dist as rdist
    cfg, fam = _cfg(api.default_config, B=3)
    (hx, hy), centres, ids, sp = _scene(cfg)
    det = api.Detector(cfg)
    n = 3
    poses = np.concatenate([synth.sample_poses(1, cfg, seed=300 + f, z_range=(0.9, 1.6), max_tilt_deg=40, half_extent_m=(hx, hy)) for f in range(n)])
    frames = torch.empty((n, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
    det.synth_render(sp, poses, frames)
    frames[1].fill_(128)                                    # a frame without tags
    torch.cuda.synchronize()
    nslots = det.record_slots(n)
    assert nslots == n * cfg.max_targets
    tab = torch.full((nslots, rdist.REC), -1.0, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    det.set_record_tables(tab, None, frame_offset=50)
    dets, _ = det.detect(frames, n)
    assert len(dets) == 2 * GX * GY and set(dets.frame.tolist()) == {0, 2}
    got = tab.cpu().numpy()
    assert np.array_equal(got, rdist.pack(dets, n, cfg.max_targets, 50))
    assert (got[cfg.max_targets:2 * cfg.max_targets] == 0).all()
    with pytest.raises(ValueError):
        rdist.pack(dets, n, GX * GY - 1, 50)               # too few slots per frame: an error, never a cut
    det.close()


@pytest.mark.gpu
@pytest.mark.parametrize("reference_mode", [0, 1])
def test_tag_pose_mfma_form_matches_vector_form_and_oracle(oracle, reference_mode):
    """rcc_config.pnp_use_mfma (BASELINE.json north_star: MFMA for the batched JtJ / Jtr blocks; configs[4]): the 4-point
    tag poses with their normal equations accumulated by v_mfma_f64_16x16x4_f64, two targets per instruction, against the
    vector form (default) and against the oracle (bar 1e-4) -- sub-pixel corners and int-truncated ones
    (corner_detections.cpp:53-54).  The flag is honoured both from the configuration and from rcc_set_pnp_mfma."""
    import torch
    cfg, fam = _cfg(api.default_config, B=3)
    cfg.reference_mode = reference_mode
    (hx, hy), centres, ids, sp = _scene(cfg)
    n = 3
    poses = np.concatenate([synth.sample_poses(1, cfg, seed=400 + f, z_range=(0.9, 1.6), max_tilt_deg=40, half_extent_m=(hx, hy)) for f in range(n)])
    det = api.Detector(cfg)
    frames = torch.empty((n, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
    det.synth_render(sp, poses, frames)
    torch.cuda.synchronize()
    v, _ = det.detect(frames, n)
    assert det.set_pnp_mfma(1) == 0
    m, _ = det.detect(frames, n)
    assert det.set_pnp_mfma(0) == 1
    det.close()
    cfg2 = api.clone_config(cfg)
    cfg2.pnp_use_mfma = 1
    det2 = api.Detector(cfg2)
    m2, _ = det2.detect(frames, n)
    det2.close()
    assert len(v) == len(m) == len(m2) == n * GX * GY
    assert (v.id == m.id).all() and (v.frame == m.frame).all() and (v.pnp_status == m.pnp_status).all() and (m.pnp_status == 0).all()
    assert m.tobytes() == m2.tobytes()
    assert np.abs(v.rvec - m.rvec).max() <= 1e-6 and np.abs(v.tvec - m.tvec).max() <= 1e-6 and np.abs(v.rms - m.rms).max() <= 1e-6   # measured 1.6e-8 over 24 456 tags; bar 1e-4
    assert (v.pnp_iters == m.pnp_iters).mean() >= 0.95
    host = frames.cpu().numpy()
    ctx = oracle.Context(cfg)
    k0 = 0
    for f in range(n):
        k, odet, ofc = ctx.detect(host[f], f)
        for q in range(k):
            a, b = m[k0 + q], odet[q]
            assert a.id == b.id
            assert np.abs(a.rvec - np.array(list(b.rvec))).max() <= 1e-4 and np.abs(a.tvec - np.array(list(b.tvec))).max() <= 1e-4
        k0 += k


@pytest.mark.gpu
def test_shim_configuration_end_to_end_through_the_reference_consumer():
    """The optional ROS node (host/tag_detections_shim.cpp) configured as it configures itself -- RAW image
    (undistort = 0), square fiducials, plumb-bob D -- and then the reference's own chain replayed on its output:
    message fields (host/tag_detections_fill.h) -> the consumer's int() casts and detections_N.yaml
    (corner_detections.cpp:46-56, 27-37) -> camera_pose_node's solve: object points +-size/2, the int corners, K and D
    (camera_pose.cpp:152-163, through rcc_solve_pnp_batch).  The poses must be the rendered ones up to the pixel
    truncation the reference applies itself -- which they are not if the corners were those of an undistorted image
    (distortion applied twice) or if `size` did not span the four corners."""
    import ctypes as C, os, yaml
    import torch
    import tests.test_tagmap_yaml as TY
    cfg, fam = _cfg(api.default_config, B=2)
    cfg.undistort = 0                                       # as the node sets it: corners of the raw image
    (hx, hy), centres, ids, sp = _scene(cfg)
    det = api.Detector(cfg)
    n = 2
    poses = np.concatenate([synth.sample_poses(1, cfg, seed=500 + f, z_range=(0.9, 1.4), max_tilt_deg=35, half_extent_m=(hx, hy)) for f in range(n)])
    frames = torch.empty((n, cfg.frame_bytes), dtype=torch.uint8, device="cuda:0")
    det.synth_render(sp, poses, frames)
    dets, _ = det.detect(frames, n)
    assert len(dets) == n * GX * GY
    TY.test_ros_shim_message_filling()                      # (re)builds the message-filling library
    Ls = C.CDLL(os.path.join(TY.ROOT, "tests", "host", "libshimfill_host.so"))
    Lt = C.CDLL(TY.SO)
    Lt.rcc_yaml_detections.restype = C.c_size_t
    Lt.rcc_yaml_detections.argtypes = [C.c_char_p, C.c_size_t, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    K = np.array(list(cfg.K)); D = np.array(list(cfg.D))
    worst_t = worst_R = 0.0
    for f in range(n):
        mine = dets[dets.frame == f]
        m = len(mine)
        recs = (abi.rcc_detection * m).from_buffer_copy(np.ascontiguousarray(mine).tobytes())
        out = np.zeros(13 * m); asint = np.zeros(8 * m, np.int32)
        assert Ls.shimfill_roundtrip(recs, m, f, out.ctypes.data_as(C.c_void_p), asint.ctypes.data_as(C.c_void_p)) == m
        ids_m = np.ascontiguousarray(out.reshape(m, 13)[:, 0].astype(np.int32))
        sizes = np.ascontiguousarray(out.reshape(m, 13)[:, 1])
        corners = np.ascontiguousarray(np.stack([asint.reshape(m, 2, 4)[:, 0, :], asint.reshape(m, 2, 4)[:, 1, :]], -1).astype(np.int32))   # m x 4 x (x, y)
        buf = C.create_string_buffer(1 << 16)
        Lt.rcc_yaml_detections(buf, len(buf), m, ids_m.ctypes.data_as(C.c_void_p), sizes.ctypes.data_as(C.c_void_p), corners.ctypes.data_as(C.c_void_p))
        doc = yaml.safe_load(buf.value.decode())            # what camera_pose.cpp:134-143 reads back
        objs, imgs, tag_ids = [], [], []
        for d in doc["detections"]:
            s = float(d["size"][0]) / 2.0
            objs.append(np.array([[-s, -s, 0], [s, -s, 0], [s, s, 0], [-s, s, 0]]))          # camera_pose.cpp:158-161
            imgs.append(np.array([d["corners"][k] for k in range(4)], float))                # bl, br, tr, tl (:152-155)
            tag_ids.append(int(d["targetID"]))
        rv, tv, rms, st, it = det.solve_pnp(objs, imgs, K, D, abi.RCC_DIST_PLUMB_BOB)         # solvePnP(obj, img, K, D, ...) (:163)
        assert (st == 0).all() and sorted(tag_ids) == list(ids)
        R = synth.rodrigues(poses[f][:3])
        for q, i in enumerate(tag_ids):
            c = centres[list(ids).index(i)]
            worst_t = max(worst_t, np.abs(tv[q] - (R @ c + poses[f][3:])).max())
            worst_R = max(worst_R, np.abs(synth.rodrigues(rv[q]) - R).max())
    det.close()
    assert worst_t < 0.04 and worst_R < 0.2, (worst_t, worst_R)      # 4-point poses from int-truncated corners of ~60 px tags
