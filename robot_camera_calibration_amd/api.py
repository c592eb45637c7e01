"""Host-side mirror of the reference's interface for the detection + pose path, over the C ABI.

The reference exposes this path as (i) a ROS topic carrying, per detection, id[0], size[0] and
pixel_corners_x/y[0..3] (real_preprocessing/src/corner_detections.cpp:41-56,78) and (ii) one call
cv::solvePnP(obj_pts, img_pts, K, D, rvec, tvec, false, CV_ITERATIVE) followed by cv::Rodrigues
(real_preprocessing/src/camera_pose.cpp:163-164).  `Detector.detect` returns records with exactly
those fields (+ pose); `Detector.solve_pnp` takes the arguments of the solver call in the same
order and meaning.  Everything computes in librcc_hip.so (hand-written HIP, gfx950); there is no
CPU fallback -- a missing library or device raises.
"""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBNAME = "librcc_hip.so"
_lib = None


class RccError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        msg = "%s failed: %s (%d)" % (where, status_string(status), status)
        if detail:
            msg += " -- " + detail
        super().__init__(msg)


def library_path():
    """RCC_LIBRARY overrides the in-tree build (A/B of two builds of the same source); it must
    still be a build of this package's csrc/ -- there is no other implementation to point it at."""
    return os.environ.get("RCC_LIBRARY") or os.path.join(_HERE, _LIBNAME)


def load_library():
    """Load the HIP implementation.  Raises if it has not been built (python __graft_entry__.py
    build): the product path never substitutes anything for it."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise FileNotFoundError(
            "%s not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()')" % path)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64, and whichever copy is loaded first serves
    # both.  With the system copy first, torch finds no device afterwards ("No HIP GPUs are available"), so where torch is
    # installed it is imported before this library is loaded (callers that never use torch lose ~1.5 s once).
    if "torch" not in sys.modules and importlib.util.find_spec("torch") is not None:
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(path)
    P, I = C.c_void_p, C.c_int32
    L.rcc_abi_version.restype = C.c_int
    L.rcc_status_string.restype = C.c_char_p
    L.rcc_status_string.argtypes = [C.c_int]
    L.rcc_last_device_error.restype = C.c_char_p
    L.rcc_last_device_error.argtypes = [P]
    L.rcc_default_config.argtypes = [C.POINTER(abi.rcc_config)]
    L.rcc_default_config.restype = None
    L.rcc_create.argtypes = [C.POINTER(abi.rcc_config), C.POINTER(P)]
    L.rcc_destroy.argtypes = [P]
    L.rcc_destroy.restype = None
    L.rcc_detect_batch.argtypes = [P, P, I, I, P, C.POINTER(I), P, P]
    L.rcc_detect_batch_submit.argtypes = [P, P, I, I, P, P]
    L.rcc_detect_batch_submit.restype = C.c_int
    L.rcc_detect_batch_collect.argtypes = [P, P, C.POINTER(I)]
    L.rcc_detect_batch_collect.restype = C.c_int
    L.rcc_solve_pnp_batch.argtypes = [P, P, P, P, I, P, P, I, P, P, P, P, P]
    L.rcc_rodrigues_v2m_batch.argtypes = [P, P, I, P]
    L.rcc_rodrigues_m2v_batch.argtypes = [P, P, I, P]
    L.rcc_stage_ingest.argtypes = [P, P, I, P, P]
    L.rcc_stage_threshold_corner.argtypes = [P, P, I, P, P, P, P]
    L.rcc_stage_targets.argtypes = [P, P, P, P, P, I, P, C.POINTER(I), P, P]
    L.rcc_set_dense_variant.argtypes = [P, C.c_int]
    L.rcc_set_ingest_variant.argtypes = [P, C.c_int]
    L.rcc_set_pnp_variant.argtypes = [P, C.c_int]
    L.rcc_set_dense_skip.argtypes = [P, C.c_int]
    L.rcc_set_dense_skip.restype = C.c_int
    L.rcc_time_copy.argtypes = [P, P, P, C.c_int64, I, C.POINTER(C.c_float)]
    L.rcc_time_copy.restype = C.c_int
    L.rcc_set_fuse_grid_pnp.argtypes = [P, C.c_int]
    L.rcc_set_fuse_grid_pnp.restype = C.c_int
    L.rcc_set_keep_binary.argtypes = [P, C.c_int]
    L.rcc_set_keep_binary.restype = C.c_int
    if hasattr(L, "rcc_set_dense_gang"):          # librcc_hip_exp.so only (include/rcc_debug.h, RCC_EXPERIMENTS)
        L.rcc_set_dense_gang.argtypes = [P, C.c_int, C.c_int]
        L.rcc_set_dense_gang.restype = C.c_int
    L.rcc_set_host_chunk.argtypes = [P, C.c_int]
    L.rcc_set_host_chunk.restype = C.c_int
    L.rcc_set_subpix_grid.argtypes = [P, C.c_int]
    L.rcc_set_subpix_grid.restype = C.c_int
    L.rcc_set_pipeline.argtypes = [P, C.c_int]
    L.rcc_set_pipeline.restype = C.c_int
    L.rcc_set_pnp_variant.restype = C.c_int
    L.rcc_last_timings.argtypes = [P, P, I]
    L.rcc_last_step_times.argtypes = [P, P, I]
    L.rcc_last_step_times.restype = C.c_int
    L.rcc_debug_measure_clock.argtypes = [P, I, C.c_float, P]
    L.rcc_debug_measure_clock.restype = C.c_int
    L.rcc_time_dense.argtypes = [P, P, I, P, P, P, I, C.POINTER(C.c_float)]
    L.rcc_time_ingest.argtypes = [P, P, I, P, I, C.POINTER(C.c_float)]
    L.rcc_synth_render_batch.argtypes = [P, C.POINTER(abi.rcc_synth_params), P, I, I, P, P]
    L.rcc_debug_fetch_lists.argtypes = [P, I, P, P, P, P, P]
    L.rcc_debug_fetch_images.argtypes = [P, I, P, P, P, P]
    L.rcc_debug_calib_copy.argtypes = [P, P, P, C.c_int64]
    L.rcc_debug_calib_copy.restype = C.c_int
    L.rcc_last_dense_kernel.argtypes = [P]
    L.rcc_last_dense_kernel.restype = C.c_char_p
    L.rcc_set_pnp_mfma.argtypes = [P, C.c_int]
    L.rcc_set_pnp_mfma.restype = C.c_int
    L.rcc_set_record_tables.argtypes = [P, P, P, I, I]
    L.rcc_set_record_tables.restype = C.c_int
    L.rcc_record_slots.argtypes = [P, I]
    L.rcc_record_slots.restype = C.c_int
    for name in ("rcc_create", "rcc_detect_batch", "rcc_solve_pnp_batch", "rcc_rodrigues_v2m_batch",
                 "rcc_rodrigues_m2v_batch", "rcc_stage_ingest", "rcc_stage_threshold_corner", "rcc_stage_targets",
                 "rcc_set_dense_variant", "rcc_set_ingest_variant", "rcc_last_timings", "rcc_time_dense",
                 "rcc_time_ingest", "rcc_synth_render_batch", "rcc_debug_fetch_lists", "rcc_debug_fetch_images"):
        getattr(L, name).restype = C.c_int
    if L.rcc_abi_version() != abi.RCC_ABI_VERSION:
        raise RuntimeError("librcc_hip.so ABI version %d != %d" % (L.rcc_abi_version(), abi.RCC_ABI_VERSION))
    _lib = L
    return L


# every symbol include/rcc.h declares -- the drop-in boundary (the CPU-side test checks the library exports them all)
EXPORTED_SYMBOLS = (
    "rcc_default_config", "rcc_create", "rcc_destroy", "rcc_status_string", "rcc_last_device_error",
    "rcc_abi_version", "rcc_detect_batch", "rcc_detect_batch_submit", "rcc_detect_batch_collect",
    "rcc_solve_pnp_batch", "rcc_rodrigues_v2m_batch", "rcc_rodrigues_m2v_batch",
    "rcc_stage_ingest", "rcc_stage_threshold_corner", "rcc_stage_targets",
    "rcc_set_keep_binary", "rcc_set_pnp_mfma", "rcc_set_record_tables", "rcc_record_slots", "rcc_synth_render_batch",
)
# include/rcc_debug.h: test taps, timers, A/B switches between bit-identical variants (not part of the boundary)
DEBUG_EXPORTED_SYMBOLS = (
    "rcc_set_dense_variant", "rcc_set_ingest_variant", "rcc_set_dense_skip", "rcc_set_fuse_grid_pnp", "rcc_set_pipeline", "rcc_set_host_chunk", "rcc_set_subpix_grid",
    "rcc_set_pnp_variant", "rcc_last_timings", "rcc_last_step_times", "rcc_debug_measure_clock", "rcc_last_dense_kernel", "rcc_time_dense", "rcc_time_ingest", "rcc_time_copy",
    "rcc_debug_calib_copy", "rcc_debug_fetch_lists", "rcc_debug_fetch_images", "rcc_debug_pnp_probe",
)
# only in librcc_hip_exp.so (make EXPERIMENTS=1): measurement-only kernel forms and one-off experiments; never in the product library
EXPERIMENT_ONLY_SYMBOLS = ("rcc_set_dense_gang", "rcc_debug_overlap", "rcc_set_dense_fmod", "rcc_debug_grid_trace", "rcc_set_tail_overlap")
# include/rcc_dist.h (librcc_dist.so: the RCCL all-gather of the record tables for non-Python hosts)
DIST_EXPORTED_SYMBOLS = ("rcc_dist_unique_id", "rcc_dist_create", "rcc_dist_destroy", "rcc_dist_rank", "rcc_dist_world",
                         "rcc_dist_allgather_records", "rcc_dist_last_error", "rcc_dist_last_create_error")


def dist_library_path():
    return os.path.join(_HERE, "librcc_dist.so")


def status_string(status):
    try:
        return load_library().rcc_status_string(int(status)).decode()
    except Exception:  # library absent: still give a readable message
        return {0: "ok", -1: "invalid argument", -2: "unsupported configuration", -3: "HIP device error",
                -4: "batch exceeds handle capacity", -5: "out of memory"}.get(int(status), "unknown status")


def default_config():
    cfg = abi.rcc_config()
    load_library().rcc_default_config(C.byref(cfg))
    return cfg


def clone_config(cfg):
    """a byte copy of an rcc_config (pointers inside -- the family table -- are shared)"""
    c = abi.rcc_config()
    C.memmove(C.byref(c), C.byref(cfg), C.sizeof(abi.rcc_config))
    return c


def _ptr(x):
    """Raw address of a numpy array, a torch tensor (host or device), an int, or None."""
    if x is None:
        return None
    if isinstance(x, int):
        return C.c_void_p(x)
    if isinstance(x, np.ndarray):
        return C.c_void_p(x.ctypes.data)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    raise TypeError("cannot take the address of %r" % type(x))


def _is_device(x):
    return hasattr(x, "is_cuda") and bool(x.is_cuda)


def _after_torch(stream, *args):
    """With stream=None the library works on the handle's own non-blocking stream, which is not ordered after torch's
    current stream: wait for that stream first if any argument is a torch device tensor (a no-op when it is idle)."""
    if stream is not None:
        return
    for x in args:
        if _is_device(x):
            import torch
            torch.cuda.current_stream(x.device).synchronize()
            return


CAND_DT = np.dtype([("x", np.int16), ("y", np.int16), ("score", np.int32)])
# numpy mirrors of the result records (same layout as abi.rcc_detection / abi.rcc_frame_corners)
DET_DT = np.dtype([("frame", "<i4"), ("id", "<i4"), ("hamming", "<i4"), ("ncorners", "<i4"), ("size", "<f8"),
                   ("corners", "<f8", (4, 2)), ("rvec", "<f8", (3,)), ("tvec", "<f8", (3,)), ("rms", "<f8"),
                   ("pnp_status", "<i4"), ("pnp_iters", "<i4")])
FC_DT = np.dtype([("status", "<i4"), ("ncand", "<i4"), ("nkept", "<i4"), ("ncorners", "<i4"),
                  ("px", "<i4", (abi.RCC_MAX_BOARD_CORNERS, 2)), ("xy", "<f8", (abi.RCC_MAX_BOARD_CORNERS, 2))])
assert DET_DT.itemsize == C.sizeof(abi.rcc_detection) and FC_DT.itemsize == C.sizeof(abi.rcc_frame_corners)


class Detector:
    """One handle = one GPU + one configuration (not thread-safe; one per thread)."""

    def __init__(self, cfg):
        self._L = load_library()
        self.cfg = cfg
        h = C.c_void_p()
        st = self._L.rcc_create(C.byref(cfg), C.byref(h))
        if st != abi.RCC_OK:
            raise RccError(st, "rcc_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.rcc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, st, where):
        if st != abi.RCC_OK:
            raise RccError(st, where, self._L.rcc_last_device_error(self._h).decode())

    # ---- the hot path -------------------------------------------------------------------------
    def detect(self, frames, nframes=None, want_corners=True, stream=None):
        """frames: numpy uint8 array (host) or torch uint8 tensor (host or device), nframes images
        frame_bytes apart.  Returns (detections, frame_corners): numpy record arrays with the
        layouts of rcc_detection (length ndet) and rcc_frame_corners (length nframes, or None);
        fields are read as attributes (d.id, d.size, d.corners, d.rvec ...)."""
        if nframes is None:
            nframes = int(frames.shape[0])
        mem = abi.RCC_MEM_DEVICE if _is_device(frames) else abi.RCC_MEM_HOST
        if mem == abi.RCC_MEM_HOST and isinstance(frames, np.ndarray):
            frames = np.ascontiguousarray(frames)
        det = np.zeros(max(nframes * self.cfg.max_targets, 1), DET_DT)
        fc = np.zeros(max(nframes, 1), FC_DT) if want_corners else None
        ndet = C.c_int32(0)
        _after_torch(stream, frames)
        self.last_stream = stream            # the HIP stream handle this batch ran on (None: the handle's own stream)
        st = self._L.rcc_detect_batch(self._h, _ptr(frames), nframes, mem, _ptr(det), C.byref(ndet),
                                      _ptr(fc), _ptr(stream))
        self._chk(st, "rcc_detect_batch")
        return det[:ndet.value].view(np.recarray), (fc[:nframes].view(np.recarray) if fc is not None else None)

    def submit(self, frames, nframes=None, want_corners=False, stream=None):
        """Asynchronous detect(): launch a batch and return; at most two may be outstanding.  collect() hands back
        the oldest one's results.  `frames` must stay alive (and unmodified) until that batch has been collected."""
        if nframes is None:
            nframes = int(frames.shape[0])
        mem = abi.RCC_MEM_DEVICE if _is_device(frames) else abi.RCC_MEM_HOST
        if mem == abi.RCC_MEM_HOST and isinstance(frames, np.ndarray):
            frames = np.ascontiguousarray(frames)
        fc = np.zeros(max(nframes, 1), FC_DT) if want_corners else None
        _after_torch(stream, frames)
        self.last_stream = stream
        self._chk(self._L.rcc_detect_batch_submit(self._h, _ptr(frames), nframes, mem, _ptr(fc), _ptr(stream)), "rcc_detect_batch_submit")
        if not hasattr(self, "_pending"):
            self._pending = []
            self._nsub = 0
        self._pending.append((frames, nframes, fc, self._nsub & 1))
        self._nsub += 1

    def collect(self):
        """results of the oldest outstanding submit(): (detections, frame_corners or None), as detect() returns them"""
        frames, nframes, fc, self.last_slot = self._pending.pop(0)      # last_slot: which record table holds this batch
        det = np.zeros(max(nframes * self.cfg.max_targets, 1), DET_DT)
        ndet = C.c_int32(0)
        self._chk(self._L.rcc_detect_batch_collect(self._h, _ptr(det), C.byref(ndet)), "rcc_detect_batch_collect")
        return det[:ndet.value].view(np.recarray), (fc[:nframes].view(np.recarray) if fc is not None else None)

    def solve_pnp(self, obj_pts, img_pts, K=None, D=None, dist_model=None):
        """Batched cv::solvePnP(obj, img, K, D, rvec, tvec, false, CV_ITERATIVE)
        (camera_pose.cpp:163).  obj_pts/img_pts: lists of (n_i,3)/(n_i,2) arrays, or single arrays
        of shape (T,n,3)/(T,n,2).  Returns rvec (T,3), tvec (T,3), rms (T,), status (T,), iters (T,)."""
        objs = [np.asarray(o, np.float64).reshape(-1, 3) for o in obj_pts]
        imgs = [np.asarray(i, np.float64).reshape(-1, 2) for i in img_pts]
        T = len(objs)
        npts = np.array([len(o) for o in objs], np.int32)
        obj = np.ascontiguousarray(np.concatenate(objs, 0)) if T else np.zeros((0, 3))
        img = np.ascontiguousarray(np.concatenate(imgs, 0)) if T else np.zeros((0, 2))
        rvec = np.zeros((T, 3)); tvec = np.zeros((T, 3)); rms = np.zeros(T)
        status = np.zeros(T, np.int32); iters = np.zeros(T, np.int32)
        Kp = np.ascontiguousarray(K, np.float64).reshape(9) if K is not None else None
        Dp = None
        if D is not None:
            Dp = np.zeros(8)
            Dp[:len(D)] = np.asarray(D, np.float64).ravel()[:8]
        model = self.cfg.dist_model if dist_model is None else dist_model
        st = self._L.rcc_solve_pnp_batch(self._h, _ptr(obj), _ptr(img), _ptr(npts), T, _ptr(Kp), _ptr(Dp), model,
                                         _ptr(rvec), _ptr(tvec), _ptr(rms), _ptr(status), _ptr(iters))
        self._chk(st, "rcc_solve_pnp_batch")
        return rvec, tvec, rms, status, iters

    def rodrigues(self, x):
        """cv::Rodrigues both ways (camera_pose.cpp:93,116,164): (n,3) -> (n,3,3) or (n,3,3) -> (n,3)."""
        x = np.ascontiguousarray(x, np.float64)
        if x.ndim == 2 and x.shape[1] == 3:
            out = np.zeros((len(x), 3, 3))
            self._chk(self._L.rcc_rodrigues_v2m_batch(self._h, _ptr(x), len(x), _ptr(out)), "rcc_rodrigues_v2m_batch")
            return out
        x = x.reshape(-1, 9)
        out = np.zeros((len(x), 3))
        self._chk(self._L.rcc_rodrigues_m2v_batch(self._h, _ptr(x), len(x), _ptr(out)), "rcc_rodrigues_m2v_batch")
        return out

    # ---- stage-level (device pointers) -----------------------------------------------------------
    def stage_ingest(self, d_frames, nframes, d_grey, stream=None):
        _after_torch(stream, d_frames, d_grey)
        self._chk(self._L.rcc_stage_ingest(self._h, _ptr(d_frames), nframes, _ptr(d_grey), _ptr(stream)), "rcc_stage_ingest")

    def stage_threshold_corner(self, d_grey, nframes, d_bin, d_cand, d_count, stream=None):
        _after_torch(stream, d_grey, d_bin, d_cand, d_count)
        self._chk(self._L.rcc_stage_threshold_corner(self._h, _ptr(d_grey), nframes, _ptr(d_bin), _ptr(d_cand),
                                                     _ptr(d_count), _ptr(stream)), "rcc_stage_threshold_corner")

    def stage_targets(self, d_grey, d_bin, d_cand, d_count, nframes, want_corners=True, stream=None):
        det = np.zeros(max(nframes * self.cfg.max_targets, 1), DET_DT)
        fc = np.zeros(max(nframes, 1), FC_DT) if want_corners else None
        ndet = C.c_int32(0)
        _after_torch(stream, d_grey, d_bin, d_cand, d_count)
        st = self._L.rcc_stage_targets(self._h, _ptr(d_grey), _ptr(d_bin), _ptr(d_cand), _ptr(d_count), nframes,
                                       _ptr(det), C.byref(ndet), _ptr(fc), _ptr(stream))
        self._chk(st, "rcc_stage_targets")
        return det[:ndet.value].view(np.recarray), (fc[:nframes].view(np.recarray) if fc is not None else None)

    def record_slots(self, nframes):
        """slots of the record table of an nframes batch (rcc_set_record_tables)"""
        return int(self._L.rcc_record_slots(self._h, int(nframes)))

    def set_record_tables(self, t0, t1=None, frame_offset=0, capacity_slots=None):
        """Device tables (capacity_slots x abi.RCC_REC_DOUBLES float64 each; default: t0's first dimension) that every
        following detect() / submit() also fills on the device: t0 for detect() and submissions in result slot 0, t1 for
        those in slot 1.  A batch that needs more slots is refused (RCC_ERR_CAPACITY), a shorter one leaves the rest of the
        table zero.  The tensors must stay alive while they are registered; (None, None) switches the tables off."""
        self._rec_tables = (t0, t1)
        if capacity_slots is None:
            capacity_slots = 0 if t0 is None else int(t0.shape[0])
            if t1 is not None:
                capacity_slots = min(capacity_slots, int(t1.shape[0]))
        self._chk(self._L.rcc_set_record_tables(self._h, _ptr(t0), _ptr(t1), int(capacity_slots), int(frame_offset)), "rcc_set_record_tables")

    def set_dense_variant(self, v):
        return self._L.rcc_set_dense_variant(self._h, int(v))

    def set_dense_skip(self, on):
        return self._L.rcc_set_dense_skip(self._h, int(on))

    def time_copy(self, d_src, d_dst, nbytes, reps=5):
        """mean ms of a plain streaming copy of nbytes (the yardstick of the bandwidth-bound passes)"""
        ms = C.c_float(0)
        self._chk(self._L.rcc_time_copy(self._h, _ptr(d_src), _ptr(d_dst), int(nbytes), int(reps), C.byref(ms)), "rcc_time_copy")
        return float(ms.value)

    def set_fuse_grid_pnp(self, on):
        return self._L.rcc_set_fuse_grid_pnp(self._h, int(on))

    def set_keep_binary(self, on):
        """1: detect() materialises the full binary image; 0 (default): the compact per-tile threshold map"""
        return self._L.rcc_set_keep_binary(self._h, int(on))

    def set_pipeline(self, nchunks):
        """chunks of detect()'s two-stream pipeline (0/1: single pass, per-stage timings available)"""
        return self._L.rcc_set_pipeline(self._h, int(nchunks))

    def set_dense_gang(self, sync_rows, segments=0):
        """experiments library only (RCC_LIBRARY=.../librcc_hip_exp.so): k_dense_wave as gangs of eight windows"""
        if not hasattr(self._L, "rcc_set_dense_gang"):
            raise RuntimeError("rcc_set_dense_gang exists only in librcc_hip_exp.so (make -C csrc EXPERIMENTS=1; RCC_LIBRARY selects it)")
        return self._L.rcc_set_dense_gang(self._h, int(sync_rows), int(segments))

    def set_tail_overlap(self, mode):
        """experiments library only: streamed board batches run their lattice + pose kernel on a stream of its own, under the next
        batch's ingest pass (1 on, 0 off)"""
        if not hasattr(self._L, "rcc_set_tail_overlap"):
            raise RuntimeError("rcc_set_tail_overlap exists only in librcc_hip_exp.so")
        self._L.rcc_set_tail_overlap.argtypes = [C.c_void_p, C.c_int]
        self._L.rcc_set_tail_overlap.restype = C.c_int
        return self._chk(self._L.rcc_set_tail_overlap(self._h, int(mode)), "rcc_set_tail_overlap")

    def set_host_chunk(self, frames_per_chunk):
        """host-resident input: frames per chunk of the copy / compute pipeline (0 automatic, < 0 one copy then the kernels)"""
        return self._L.rcc_set_host_chunk(self._h, int(frames_per_chunk))

    def set_subpix_grid(self, width):
        """tag scenes: width of the sub-pixel kernel's grid (waves walk the candidate list with this stride; 0 automatic)"""
        return self._L.rcc_set_subpix_grid(self._h, int(width))

    def set_pnp_mfma(self, on):
        """1: 4-point tag poses accumulate their normal equations on the matrix cores (cfg.pnp_use_mfma); returns the previous setting"""
        return self._L.rcc_set_pnp_mfma(self._h, int(on))

    def set_pnp_variant(self, v):
        return self._L.rcc_set_pnp_variant(self._h, int(v))

    def set_ingest_variant(self, v):
        return self._L.rcc_set_ingest_variant(self._h, int(v))

    def last_timings(self):
        ms = (C.c_float * 5)()
        self._L.rcc_last_timings(self._h, ms, 5)
        return dict(zip(("ingest", "dense", "list_subpix_grid", "pnp", "d2h"), [float(v) for v in ms]))

    def last_step_times(self):
        """the batch collect() just handed back, ms: device time (stream reaches the batch -> records in pinned memory), device
        idle time in front of it (-1: unknown), and the five stage times as they ran inside that streamed step"""
        ms = (C.c_float * 7)()
        self._L.rcc_last_step_times(self._h, ms, 7)
        return dict(zip(("device", "idle_before", "ingest", "dense", "list_subpix_grid", "pnp", "d2h"), [float(v) for v in ms]))

    def measure_clock(self, waves_per_simd=6, ms_target=20.0):
        """engine clock held under a vector-issue load and the cost of a vector wave-instruction, measured now (k_probe.hip)"""
        out = (C.c_double * 6)()
        self._chk(self._L.rcc_debug_measure_clock(self._h, int(waves_per_simd), float(ms_target), out), "rcc_debug_measure_clock")
        return dict(clock_mhz=out[0], ns_per_wave_inst_per_simd=out[1], cycles_per_wave_inst_per_simd=out[2], probe_ms=out[3],
                    clock_mhz_min=out[4], clock_mhz_max=out[5], waves_per_simd=int(waves_per_simd))

    def last_dense_kernel(self):
        return self._L.rcc_last_dense_kernel(self._h).decode()

    def time_dense(self, d_grey, nframes, d_bin, d_cand, d_count, reps):
        ms = C.c_float(0)
        _after_torch(None, d_grey, d_bin, d_cand, d_count)
        self._chk(self._L.rcc_time_dense(self._h, _ptr(d_grey), nframes, _ptr(d_bin), _ptr(d_cand), _ptr(d_count),
                                         reps, C.byref(ms)), "rcc_time_dense")
        return float(ms.value)

    def time_ingest(self, d_frames, nframes, d_grey, reps):
        ms = C.c_float(0)
        _after_torch(None, d_frames, d_grey)
        self._chk(self._L.rcc_time_ingest(self._h, _ptr(d_frames), nframes, _ptr(d_grey), reps, C.byref(ms)), "rcc_time_ingest")
        return float(ms.value)

    def synth_render(self, sp, poses, d_frames, first_index=0, stream=None):
        poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 6)
        _after_torch(stream, d_frames)
        self._chk(self._L.rcc_synth_render_batch(self._h, C.byref(sp), _ptr(poses), len(poses), first_index,
                                                 _ptr(d_frames), _ptr(stream)), "rcc_synth_render_batch")

    # ---- test taps ---------------------------------------------------------------------------------
    def fetch_lists(self, nframes):
        kc = abi.RCC_MAX_KEPT_FIDUCIAL          # capacity of the list after suppression, whatever the target (round 4)
        pre = np.zeros((nframes, kc), CAND_DT); kept = np.zeros((nframes, 256), CAND_DT)
        npre = np.zeros(nframes, np.int32)
        pre_xy = np.zeros((nframes, kc, 2)); kept_xy = np.zeros((nframes, 256, 2))
        self._chk(self._L.rcc_debug_fetch_lists(self._h, nframes, _ptr(pre), _ptr(npre), _ptr(pre_xy), _ptr(kept), _ptr(kept_xy)),
                  "rcc_debug_fetch_lists")
        return dict(pre=pre, npre=npre, pre_xy=pre_xy, kept=kept, kept_xy=kept_xy)

    def fetch_images(self, nframes):
        w, h = self.cfg.width, self.cfg.height
        grey = np.zeros((nframes, h, w), np.uint8); binm = np.zeros((nframes, h, w), np.uint8)
        cand = np.zeros((nframes, self.cfg.max_candidates), CAND_DT)
        cnt = np.zeros(nframes, np.int32)
        self._chk(self._L.rcc_debug_fetch_images(self._h, nframes, _ptr(grey), _ptr(binm), _ptr(cand), _ptr(cnt)),
                  "rcc_debug_fetch_images")
        return dict(grey=grey, bin=binm, cand=cand, cand_count=cnt)


def detections_to_dicts(dets):
    """Plain-python view of rcc_detection records with the field names the reference reads
    (corner_detections.cpp:48-54): id, size, pixel_corners_x/y (bl,br,tr,tl) + pose."""
    out = []
    for d in dets:
        out.append(dict(frame=int(d.frame), id=int(d.id), size=float(d.size),
                        pixel_corners_x=[float(d.corners[k][0]) for k in range(4)],
                        pixel_corners_y=[float(d.corners[k][1]) for k in range(4)],
                        rvec=[float(v) for v in d.rvec], tvec=[float(v) for v in d.tvec], rms=float(d.rms),
                        pnp_status=int(d.pnp_status), pnp_iters=int(d.pnp_iters), ncorners=int(d.ncorners),
                        hamming=int(d.hamming)))
    return out
