"""CPU-side checks of the drop-in boundary: struct layouts of include/rcc.h vs the ctypes/numpy
mirrors, every declared symbol exported by librcc_hip.so, host-only entry points, and that the
product fails loudly (no CPU fallback) when there is no device."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from robot_camera_calibration_amd import abi, api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(api.library_path()):
        sys.path.insert(0, ROOT)
        import __graft_entry__ as g
        g.build()
    return api.load_library()


def test_struct_layouts_match_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "rcc.h"\nint main(){'
                   'printf("%zu %zu %zu %zu ", sizeof(rcc_config), sizeof(rcc_detection), sizeof(rcc_frame_corners), sizeof(rcc_synth_params));'
                   'printf("%zu %zu %zu %zu %zu ", offsetof(rcc_config,K), offsetof(rcc_config,D), offsetof(rcc_config,subpix_eps), offsetof(rcc_config,board_square), offsetof(rcc_config,batch_capacity));'
                   'printf("%zu %zu %zu %zu ", offsetof(rcc_detection,corners), offsetof(rcc_detection,rvec), offsetof(rcc_detection,rms), offsetof(rcc_detection,pnp_iters));'
                   'printf("%zu %zu %zu", offsetof(rcc_frame_corners,px), offsetof(rcc_frame_corners,xy), offsetof(rcc_synth_params,seed));return 0;}')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    cfg, det, fc, sp = abi.rcc_config, abi.rcc_detection, abi.rcc_frame_corners, abi.rcc_synth_params
    exp = [C.sizeof(cfg), C.sizeof(det), C.sizeof(fc), C.sizeof(sp),
           cfg.K.offset, cfg.D.offset, cfg.subpix_eps.offset, cfg.board_square.offset, cfg.batch_capacity.offset,
           det.corners.offset, det.rvec.offset, det.rms.offset, det.pnp_iters.offset,
           fc.px.offset, fc.xy.offset, sp.seed.offset]
    assert got == exp
    assert api.DET_DT.itemsize == C.sizeof(det) and api.FC_DT.itemsize == C.sizeof(fc)
    assert api.DET_DT.fields["rvec"][1] == det.rvec.offset and api.FC_DT.fields["xy"][1] == fc.xy.offset


def test_library_exports_every_declared_symbol(built):
    hdr = open(os.path.join(ROOT, "include", "rcc.h")).read()
    declared = set(re.findall(r"\b(rcc_[a-z0-9_]+)\s*\(", hdr)) - {"rcc_handle"}
    assert declared == set(api.EXPORTED_SYMBOLS), declared ^ set(api.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(built, name), "librcc_hip.so does not export %s" % name
    # the drop-in header carries no tap, timer or experiment; those live in rcc_debug.h
    assert not [n for n in declared if n.startswith(("rcc_debug_", "rcc_time_"))]
    dbg = open(os.path.join(ROOT, "include", "rcc_debug.h")).read()
    product, experiments = dbg.split("#ifdef RCC_EXPERIMENTS")
    ddecl = set(re.findall(r"\b(rcc_[a-z0-9_]+)\s*\(", product))
    assert ddecl == set(api.DEBUG_EXPORTED_SYMBOLS), ddecl ^ set(api.DEBUG_EXPORTED_SYMBOLS)
    for name in ddecl:
        assert hasattr(built, name), "librcc_hip.so does not export %s" % name
    # experiment entry points are NOT in the product library, and it reads no environment variable
    for name in set(re.findall(r"\b(rcc_[a-z0-9_]+)\s*\(", experiments)):
        assert not hasattr(built, name), "%s belongs to the -DRCC_EXPERIMENTS build only" % name
    blob = open(api.library_path(), "rb").read()
    for var in (b"RCC_DENSE_MEMONLY", b"RCC_DENSE_NSEG", b"RCC_DENSE_FCHUNK", b"RCC_RUNS_NSEG", b"RCC_INGEST_FPB", b"RCC_PNP_SOLVER"):
        assert var not in blob, "the product library still knows the environment variable %s" % var.decode()
    # measurement-only kernel forms (the two-kernel threshold + corner variant, the gang form of k_dense_wave) are not shipped either
    assert set(re.findall(r"\b(rcc_[a-z0-9_]+)\s*\(", experiments)) == set(api.EXPERIMENT_ONLY_SYMBOLS)
    assert b"k_dense_runs" not in blob and b"k_mix" not in blob
    exp_path = os.path.join(os.path.dirname(api.library_path()), "librcc_hip_exp.so")
    if os.path.exists(exp_path):
        E = C.CDLL(exp_path) if "C" in globals() else __import__("ctypes").CDLL(exp_path)
        for name in api.EXPORTED_SYMBOLS + api.DEBUG_EXPORTED_SYMBOLS + api.EXPERIMENT_ONLY_SYMBOLS:
            assert hasattr(E, name), "librcc_hip_exp.so does not export %s" % name
        assert b"k_dense_runs" in open(exp_path, "rb").read()


def test_dist_library_exports_every_declared_symbol(built):
    """include/rcc_dist.h (librcc_dist.so: the RCCL all-gather of the record tables for C++ hosts): loads, exports every
    declared symbol, and validates arguments before touching a device"""
    hdr = open(os.path.join(ROOT, "include", "rcc_dist.h")).read()
    declared = set(re.findall(r"\b(rcc_dist_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(api.DIST_EXPORTED_SYMBOLS), declared ^ set(api.DIST_EXPORTED_SYMBOLS)
    assert os.path.exists(api.dist_library_path()), "librcc_dist.so has not been built"
    L = C.CDLL(api.dist_library_path())
    for name in declared:
        assert hasattr(L, name), "librcc_dist.so does not export %s" % name
    L.rcc_dist_create.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
    h = C.c_void_p()
    ident = (C.c_char * 128)()
    assert L.rcc_dist_create(0, 0, ident, 0, C.byref(h)) == abi.RCC_ERR_ARG          # world < 1
    assert L.rcc_dist_create(2, 2, ident, 0, C.byref(h)) == abi.RCC_ERR_ARG          # rank >= world
    assert L.rcc_dist_create(0, 1, None, 0, C.byref(h)) == abi.RCC_ERR_ARG
    assert L.rcc_dist_unique_id(None) == abi.RCC_ERR_ARG
    # a create that fails on the device side (no such device: this works with and without a GPU) leaves its reason where
    # the caller can read it -- there is no handle to ask
    L.rcc_dist_last_create_error.restype = C.c_char_p
    assert L.rcc_dist_create(0, 1, ident, 4096, C.byref(h)) == abi.RCC_ERR_DEVICE and not h.value
    assert b"device 4096" in L.rcc_dist_last_create_error()
    assert re.search(r"#define\s+RCC_REC_DOUBLES\s+19", open(os.path.join(ROOT, "include", "rcc.h")).read()) and abi.RCC_REC_DOUBLES == 19


def test_host_only_entry_points(built, oracle):
    assert built.rcc_abi_version() == abi.RCC_ABI_VERSION
    assert api.status_string(0) == "ok" and "capacity" in api.status_string(abi.RCC_ERR_CAPACITY)
    a, b = api.default_config(), oracle.default_config()
    for name, _ in abi.rcc_config._fields_:
        va, vb = getattr(a, name), getattr(b, name)
        if hasattr(va, "__len__"):
            assert list(va) == list(vb), name
        else:
            assert va == vb, name
    # argument validation happens before any device work
    assert built.rcc_create(None, C.byref(C.c_void_p())) == abi.RCC_ERR_ARG
    bad = api.default_config(); bad.struct_size = 12
    assert built.rcc_create(C.byref(bad), C.byref(C.c_void_p())) == abi.RCC_ERR_ARG
    fid = api.default_config(); fid.target_kind = abi.RCC_TARGET_FIDUCIAL      # no family table given
    assert built.rcc_create(C.byref(fid), C.byref(C.c_void_p())) == abi.RCC_ERR_ARG
    fe = api.default_config(); fe.dist_model = abi.RCC_DIST_FISHEYE; fe.undistort = 0
    assert built.rcc_create(C.byref(fe), C.byref(C.c_void_p())) == abi.RCC_ERR_UNSUPPORTED


def test_no_cpu_fallback_without_a_device(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(api.RccError) as e:
        api.Detector(api.default_config())
    assert e.value.status == abi.RCC_ERR_DEVICE


def test_create_refuses_a_host_built_against_another_abi_version(built):
    """ABI 2 (round 4): rcc_set_record_tables gained an argument in the middle of its list, reserved[0] became tag_refine, the
    synthetic camera's struct grew and two defaults changed -- a host compiled against the version-1 header must be refused
    before anything else is looked at (the check runs in front of the device check, so it is testable without a GPU)"""
    cfg = api.default_config()
    assert cfg.abi_version == abi.RCC_ABI_VERSION == 2 and built.rcc_abi_version() == 2
    cfg.abi_version = 1
    with pytest.raises(api.RccError) as e:
        api.Detector(cfg)
    assert e.value.status == abi.RCC_ERR_ARG
    cfg.abi_version = abi.RCC_ABI_VERSION
    cfg.pixfmt = 3                                   # beyond RCC_PIX_RGB8
    with pytest.raises(api.RccError) as e:
        api.Detector(cfg)
    assert e.value.status == abi.RCC_ERR_ARG
    d = api.default_config()
    assert (d.thr_min_contrast, d.harris_thresh) == (16, 10240)          # the round-4 defaults


def test_product_path_never_imports_the_oracle():
    """only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/"""
    pkg = os.path.join(ROOT, "robot_camera_calibration_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "orc_py" not in txt and "liborc" not in txt and "oracle/" not in txt.replace("# oracle/", ""), fn


def test_host_build_of_pnp_arithmetic_matches_oracle(built, oracle):
    """robot_camera_calibration_amd/csrc/pnp_core.h compiled for the host (the code the HIP kernels
    run per thread) against the oracle: catches logic slips without a GPU"""
    from robot_camera_calibration_amd import synth
    so = os.path.join(ROOT, "tests", "host", "libpnpcore_host.so")
    if not os.path.exists(so):
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", so,
                               os.path.join(ROOT, "tests", "host", "pnp_core_host.cpp"), "-lm"])
    L = C.CDLL(so)
    rng = np.random.default_rng(1)
    K = np.array([576., 0, 319.5, 0, 576., 239.5, 0, 0, 1.]); D = np.array([-0.28, 0.07, 2e-4, -1e-4, 0.0, 0, 0, 0])
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    worst = 0.0
    for trial in range(120):
        n = 4 if trial % 2 == 0 else 48
        s = rng.uniform(0.03, 0.1)
        obj = np.array([[-s, -s, 0], [s, -s, 0], [s, s, 0], [-s, s, 0]], float) if n == 4 else synth.board_object_points(8, 6, 0.108)
        R = synth.rodrigues([0, 0, rng.uniform(-3.1, 3.1)]) @ synth.rodrigues(np.array([np.cos(trial), np.sin(trial), 0]) * rng.uniform(0, 1.0)) @ np.diag([1., -1, -1])
        rv = synth.rotmat_to_rvec(R); tv = np.array([rng.uniform(-.3, .3), rng.uniform(-.2, .2), rng.uniform(0.6, 2.5)])
        model = abi.RCC_DIST_PLUMB_BOB if trial % 3 else abi.RCC_DIST_NONE
        img = np.ascontiguousarray(synth.project_points(obj, rv, tv, K, model, D) + rng.normal(0, 0.1, (n, 2)) * (trial % 5 == 0))
        st, r1, t1, rms1, it1 = oracle.solve_pnp(obj, img, K, model, D)
        r2 = np.empty(3); t2 = np.empty(3); rms2 = C.c_double(); it2 = C.c_int()
        st2 = L.pnpcore_solve(p(np.ascontiguousarray(obj)), p(img), n, p(K), model, p(D), p(r2), p(t2), C.byref(rms2), C.byref(it2))
        assert st == st2 and it1 == it2.value
        worst = max(worst, np.abs(r1 - r2).max(), np.abs(t1 - t2).max())
    assert worst < 1e-6      # product uses Cholesky / inverse iteration where the oracle decomposes fully
