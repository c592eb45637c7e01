"""ctypes mirror of include/rcc.h (the C-ABI boundary).

Plain data only: struct layouts, enums and argument marshalling.  No compute happens here.
The reference interface each entry point stands in for is cited in include/rcc.h
(real_preprocessing/src/corner_detections.cpp:41-56,78 and camera_pose.cpp:55-68,132-173).
"""
import ctypes as C

RCC_ABI_VERSION = 2

# status
RCC_OK, RCC_ERR_ARG, RCC_ERR_UNSUPPORTED, RCC_ERR_DEVICE, RCC_ERR_CAPACITY, RCC_ERR_NOMEM, RCC_ERR_STATE = 0, -1, -2, -3, -4, -5, -6
# enums
RCC_PIX_MONO8, RCC_PIX_BGR8, RCC_PIX_RGB8 = 0, 1, 2
RCC_DIST_NONE, RCC_DIST_PLUMB_BOB, RCC_DIST_FISHEYE = 0, 1, 2
RCC_TARGET_CHECKERBOARD, RCC_TARGET_FIDUCIAL = 0, 1
RCC_MEM_HOST, RCC_MEM_DEVICE = 0, 1
RCC_FRAME_OK, RCC_FRAME_CAND_OVERFLOW, RCC_FRAME_NOT_FOUND, RCC_FRAME_KEPT_OVERFLOW = 0, 1, 2, 4
RCC_PNP_OK, RCC_PNP_TOO_FEW, RCC_PNP_NONPLANAR, RCC_PNP_DEGENERATE = 0, 1, 2, 3
RCC_MAX_BOARD_CORNERS = 256
RCC_REC_DOUBLES = 19   # doubles per slot of the record tables ranks exchange (include/rcc.h)


class rcc_config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("abi_version", C.c_uint32),
        ("width", C.c_int32), ("height", C.c_int32), ("stride_bytes", C.c_int32), ("pixfmt", C.c_int32),
        ("frame_bytes", C.c_int64),
        ("K", C.c_double * 9),
        ("dist_model", C.c_int32), ("undistort", C.c_int32),
        ("D", C.c_double * 8),
        ("thr_min_contrast", C.c_int32),
        ("harris_thresh", C.c_int32), ("cand_margin", C.c_int32), ("max_candidates", C.c_int32),
        ("nms_radius", C.c_int32), ("xj_check", C.c_int32), ("max_kept", C.c_int32),
        ("subpix_win", C.c_int32), ("subpix_max_iter", C.c_int32),
        ("subpix_eps", C.c_double),
        ("target_kind", C.c_int32), ("board_cols", C.c_int32), ("board_rows", C.c_int32),
        ("board_square", C.c_double),
        ("board_id", C.c_int32), ("max_targets", C.c_int32),
        ("reference_mode", C.c_int32), ("pnp_use_mfma", C.c_int32),
        ("device", C.c_int32), ("batch_capacity", C.c_int32),
        ("family_n", C.c_int32), ("tag_max_hamming", C.c_int32),
        ("family_codes", C.c_void_p),
        ("tag_size", C.c_double),
        ("tag_refine", C.c_int32),
        ("reserved", C.c_int32 * 1),
    ]


class rcc_detection(C.Structure):
    _fields_ = [
        ("frame", C.c_int32), ("id", C.c_int32), ("hamming", C.c_int32), ("ncorners", C.c_int32),
        ("size", C.c_double),
        ("corners", (C.c_double * 2) * 4),
        ("rvec", C.c_double * 3), ("tvec", C.c_double * 3),
        ("rms", C.c_double),
        ("pnp_status", C.c_int32), ("pnp_iters", C.c_int32),
    ]


class rcc_frame_corners(C.Structure):
    _fields_ = [
        ("status", C.c_int32), ("ncand", C.c_int32), ("nkept", C.c_int32), ("ncorners", C.c_int32),
        ("px", (C.c_int32 * 2) * RCC_MAX_BOARD_CORNERS),
        ("xy", (C.c_double * 2) * RCC_MAX_BOARD_CORNERS),
    ]


class rcc_synth_params(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("board_cols", C.c_int32), ("board_rows", C.c_int32),
        ("board_square", C.c_double),
        ("margin_squares", C.c_int32), ("supersample", C.c_int32),
        ("noise_sigma", C.c_double),
        ("seed", C.c_uint64),
        ("black", C.c_int32), ("white", C.c_int32), ("background", C.c_int32),
        ("fid_grid_x", C.c_int32), ("fid_grid_y", C.c_int32), ("fid_gap_permille", C.c_int32),
        ("reserved", C.c_int32 * 2),
        ("blur_taps", C.c_int32 * 8),
        ("shade_x_permille", C.c_int32), ("shade_y_permille", C.c_int32),
        ("vignette_permille", C.c_int32),
        ("reserved2", C.c_int32 * 1),
    ]


class cand_entry(C.Structure):
    """one dense-pass candidate: {int16 x, int16 y, int32 score} (rcc_stage_threshold_corner)"""
    _fields_ = [("x", C.c_int16), ("y", C.c_int16), ("score", C.c_int32)]


def default_synth_params(cols=8, rows=6, square=0.108, seed=0xC0FFEE, noise=2.0, supersample=4):
    sp = rcc_synth_params()
    sp.struct_size = C.sizeof(rcc_synth_params)
    sp.board_cols, sp.board_rows, sp.board_square = cols, rows, square
    sp.margin_squares, sp.supersample, sp.noise_sigma = 1, supersample, noise
    sp.seed = seed
    sp.black, sp.white, sp.background = 20, 235, 128
    return sp


RCC_SYNTH_BLUR_TAPS = 8


def gaussian_taps(sigma):
    """Integer half-kernel of a Gaussian of standard deviation sigma (pixels) for rcc_synth_params.blur_taps: weight at
    distance k = 0..7, radius ceil(3 sigma) (at most 7), with taps[0] + 2 * sum(taps[1:]) == 256 exactly (largest-remainder
    rounding; the centre tap takes what symmetry leaves over).  sigma <= 0: no blur (all zero)."""
    import math
    taps = [0] * RCC_SYNTH_BLUR_TAPS
    if not sigma or sigma <= 0:
        return taps
    r = min(RCC_SYNTH_BLUR_TAPS - 1, int(math.ceil(3.0 * sigma)))
    w = [math.exp(-0.5 * (k / sigma) ** 2) for k in range(r + 1)]
    tot = w[0] + 2.0 * sum(w[1:])
    for k in range(1, r + 1):
        taps[k] = int(math.floor(256.0 * w[k] / tot + 0.5))
    taps[0] = 256 - 2 * sum(taps[1:])
    assert taps[0] > 0
    return taps


BLUR_3TAP = [128, 64, 0, 0, 0, 0, 0, 0]       # the 1-2-1 filter of SURVEY.md 8(d) ("optional 3-tap blur")


def set_optics(sp, blur=None, shade_x=0, shade_y=0, vignette=0):
    """blur: None / 0 (off), "3tap", a Gaussian sigma in pixels, or a list of 8 integer taps; shading in permille"""
    if blur is None or blur == 0:
        taps = [0] * RCC_SYNTH_BLUR_TAPS
    elif blur == "3tap":
        taps = BLUR_3TAP
    elif isinstance(blur, (int, float)):
        taps = gaussian_taps(float(blur))
    else:
        taps = list(blur) + [0] * (RCC_SYNTH_BLUR_TAPS - len(blur))
    for k in range(RCC_SYNTH_BLUR_TAPS):
        sp.blur_taps[k] = int(taps[k])
    sp.shade_x_permille, sp.shade_y_permille, sp.vignette_permille = int(shade_x), int(shade_y), int(vignette)
    return sp


def set_geometry(cfg, width, height, pixfmt=RCC_PIX_BGR8):
    """Fill the image geometry and the SURVEY 8(d) synthetic intrinsics (fx=fy=0.9 W, centre)."""
    ch = 1 if pixfmt == RCC_PIX_MONO8 else 3
    cfg.width, cfg.height, cfg.pixfmt = width, height, pixfmt
    cfg.stride_bytes = width * ch
    cfg.frame_bytes = width * ch * height
    for i in range(9):
        cfg.K[i] = 0.0
    cfg.K[0] = cfg.K[4] = 0.9 * width
    cfg.K[2] = (width - 1) * 0.5
    cfg.K[5] = (height - 1) * 0.5
    cfg.K[8] = 1.0
    return cfg


def set_distortion(cfg, model, coeffs):
    cfg.dist_model = model
    for i in range(8):
        cfg.D[i] = coeffs[i] if i < len(coeffs) else 0.0
    return cfg


PLUMB_BOB_DEFAULT = (-0.28, 0.07, 2e-4, -1e-4, 0.0)
FISHEYE_DEFAULT = (-0.02, 0.005, -0.001, 0.0002)


RCC_MAX_KEPT_FIDUCIAL = 2048
RCC_TAG_REFINE_EDGES, RCC_TAG_REFINE_CORNER_SUBPIX = 0, 1
_FAMILY_CACHE = {}


def family_path(name="family36b"):
    import os
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", name + ".txt")


def load_family(name="family36b"):
    """The build-generated 36-bit family (data/family36b.txt, made by data/make_family.py): numpy
    uint64 array.  Keep the array alive while a config points at it."""
    import os
    import numpy as np
    if name not in _FAMILY_CACHE:
        path = family_path(name)
        _FAMILY_CACHE[name] = np.array([int(l, 16) for l in open(path) if l.strip() and not l.startswith("#")], dtype=np.uint64)
    return _FAMILY_CACHE[name]


def set_fiducial_target(cfg, family, tag_size=0.10, max_hamming=2, max_targets=64, max_kept=2048):
    cfg.target_kind = RCC_TARGET_FIDUCIAL
    cfg.family_n = len(family)
    cfg.family_codes = family.ctypes.data
    cfg.tag_size = tag_size
    cfg.tag_max_hamming = max_hamming
    cfg.max_targets = max_targets
    cfg.max_kept = max_kept
    cfg.max_candidates = 4096
    cfg.xj_check = 0
    return cfg
