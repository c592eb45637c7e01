#!/usr/bin/env python3
"""bench.py -- calibration frames/s at 1920x1080 on N MI355X (BASELINE.json metric).

A step = one pass of the whole hot path (ingest -> threshold+corner -> list/sub-pixel/board
indexing -> PnP -> result D2H) over one batch of 1024 synthetic 1920x1080 BGR8 checkerboard frames
that are already resident in HBM (BASELINE.json configs[1]).  With N ranks every rank owns its own
1024-frame batch (frames are independent units: weak scaling, no data-path collective) and the
per-frame pose records are all-gathered over RCCL once per step, from a table the
detector packs on the device.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (a child
`python -m torch.distributed.run`, before anything touches a GPU) and exits with the child's code; launched by
torchrun it reads RANK / LOCAL_RANK / WORLD_SIZE.  `n_gpus` in the output is the world size RCCL reports; a world that
does not match --gpus, or fewer visible devices than ranks, is an error, never a silent single-GPU run.

Prints ONE JSON line on rank 0; see DESIGN.md section 6 for the fields.
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PCIE_GBS = 63.0                # MI355X_MICROARCH.md: PCIe Gen5 x16 (spec)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)     # the first ~5 steps after start-up run 2-10 % slower than the steady state
    ap.add_argument("--warmup", type=int, default=20)    # (scratch/t_steps.py): defaults long enough to measure the latter
    ap.add_argument("--batch", type=int, default=1024, help="frames per rank per step")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--cpu-sample", type=int, default=0, help="frames timed on the CPU oracle (rank 0, N=1); 0 = 8 per thread")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the host-input (H2D) and single-frame latency legs")
    ap.add_argument("--dense-variant", type=int, default=-1)
    ap.add_argument("--ingest-variant", type=int, default=-1)
    ap.add_argument("--roofline-reps", type=int, default=5)
    ap.add_argument("--sync-steps", action="store_true", help="time K synchronous detect() calls instead of the submit/collect stream of K batches")
    ap.add_argument("--keep-gc", action="store_true", help="A/B: leave Python's cyclic garbage collector running inside the timed regions")
    ap.add_argument("--tail-overlap", type=int, default=0, help="experiments library: lattice + pose kernel of a streamed batch under the next batch's ingest pass (1)")
    ap.add_argument("--sync-first", action="store_true", help="A/B of the order: run the synchronous comparison region BEFORE the streamed region that defines `value`")
    ap.add_argument("--pipeline", type=int, default=1, help="chunks of the detector's two-stream pipeline per step (1: single pass)")
    ap.add_argument("--fiducials", default="", help="BASELINE.json configs[4]-style run: GXxGY planar grid of square fiducials per frame, e.g. 6x4")
    ap.add_argument("--gather-side-stream", action="store_true", help="A/B: the per-step all_gather on a side stream instead of in line behind the batch")
    ap.add_argument("--no-pin", action="store_true", help="do not bind the rank's host threads to its GPU's NUMA node")
    ap.add_argument("--tag-refine", default="edges", choices=["edges", "subpix"], help="fiducial corner refinement: refine_edges form (default) or the cornerSubPix form")
    ap.add_argument("--min-contrast", type=int, default=-1, help="cfg.thr_min_contrast (-1: the library's default)")
    ap.add_argument("--harris-thresh", type=int, default=-1, help="cfg.harris_thresh (-1: the library's default)")
    ap.add_argument("--noise", type=float, default=-1.0, help="synthetic camera: sensor noise sigma in LSB per channel (default: the generator's 2)")
    ap.add_argument("--blur", default="", help="synthetic camera optics: '', '3tap' or a Gaussian sigma in pixels (rcc_synth_params.blur_taps)")
    ap.add_argument("--shade", default="", help="synthetic camera optics: 'gx,gy,vignette' in permille, e.g. 300,-200,400")
    ap.add_argument("--fisheye", action="store_true", help="BASELINE.json configs[3]-style run: fisheye model (use with --width 3840 --height 2160 --batch 256)")
    return ap.parse_args(argv)


def visible_gpus_sysfs():
    """GPUs this process could open, counted WITHOUT any HIP / torch call (the launcher parent must provably never touch
    a GPU): KFD topology nodes that have SIMDs and whose render node exists in this container, capped by
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES.  No KFD topology = no AMD GPU driver = 0."""
    import glob
    n = 0
    for props in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        kv = {}
        try:
            for line in open(props):
                parts = line.split()
                if len(parts) == 2:
                    kv[parts[0]] = parts[1]
        except OSError:
            continue
        if int(kv.get("simd_count", "0")) > 0:
            minor = kv.get("drm_render_minor")
            if minor is None or int(minor) <= 0 or os.path.exists("/dev/dri/renderD%s" % minor):
                n += 1
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def pin_to_gpu_numa(torch, local):
    """Bind this rank's host threads to the CPUs of the NUMA node its GPU hangs off (the PCI device's local_cpulist),
    intersected with the CPUs the process may use.  Silently does nothing where sysfs does not say (returns None)."""
    try:
        pr = torch.cuda.get_device_properties(local)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        base = "/sys/bus/pci/devices/" + bdf
        node = int(open(base + "/numa_node").read())
        cpus = set()
        for part in open(base + "/local_cpulist").read().strip().split(","):
            if part:
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
        allowed = os.sched_getaffinity(0)
        use = cpus & allowed
        if node < 0 or not use:
            return None
        os.sched_setaffinity(0, use)
        return {"pci": bdf, "numa_node": node, "cpus": len(use)}
    except Exception:
        return None


def sysfs_clocks(torch, local):
    """current levels of the GPU's sysfs clock tables (pp_dpm_sclk / pp_dpm_mclk of the PCI device), where readable.  The engine
    level is whatever the device idles at between two launches when the file is read -- the clock held INSIDE a kernel is the
    issue probe's (out["clock"]["clock_mhz"]); the memory clock has one level on this part."""
    out = {}
    try:
        pr = torch.cuda.get_device_properties(local)
        base = "/sys/bus/pci/devices/%04x:%02x:%02x.0/" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        for key, name in (("sclk", "pp_dpm_sclk"), ("mclk", "pp_dpm_mclk")):
            try:
                lines = [l.strip() for l in open(base + name)]
            except OSError:
                continue
            cur = [l for l in lines if l.endswith("*")]
            out[key + "_levels"] = lines
            if cur:
                out[key + "_mhz_current"] = float("".join(ch for ch in cur[0].split(":")[1] if ch.isdigit() or ch == "."))
    except Exception:
        pass
    return out


class gc_watch:
    """Python's cyclic garbage collector inside a timed region: every collection that runs is recorded (generation, milliseconds) --
    a full collection over the heap of a process that has imported torch takes milliseconds, i.e. whole steps.  Unless --keep-gc, the
    collector is run once and switched off for the region (reference counting still frees what the loop allocates)."""
    events = []
    _t0 = None

    @staticmethod
    def _cb(phase, info):
        if phase == "start":
            gc_watch._t0 = time.perf_counter()
        elif gc_watch._t0 is not None:
            gc_watch.events.append((int(info.get("generation", -1)), round(1e3 * (time.perf_counter() - gc_watch._t0), 3)))

    def __init__(self, keep):
        self.keep = keep

    def __enter__(self):
        import gc
        if not self.keep:
            gc.disable()
        gc_watch.events = []
        gc.callbacks.append(gc_watch._cb)
        return self

    def __exit__(self, *exc):
        import gc
        gc.callbacks.remove(gc_watch._cb)
        if not self.keep:
            gc.enable()


def host_counters():
    """what can stall the submitting thread without the device having anything to do with it: involuntary context switches of
    this thread, and the cgroup's CPU-quota throttling counters (cpu.stat: nr_throttled, throttled_usec)"""
    out = {}
    try:
        import resource
        ru = resource.getrusage(getattr(resource, "RUSAGE_THREAD", resource.RUSAGE_SELF))
        out["involuntary_ctx_switches"], out["voluntary_ctx_switches"] = ru.ru_nivcsw, ru.ru_nvcsw
    except Exception:
        pass
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            k, _, v = line.partition(" ")
            if k in ("nr_throttled", "throttled_usec", "nr_periods"):
                out["cgroup_" + k] = int(v)
    except Exception:
        pass
    try:
        out["cpu"] = os.sched_getcpu()
    except Exception:
        pass
    return out


def rank_report(dist, dev, world, dt_local, steps, found_local, gather, pinned):
    """What makes a sub-linear N > 1 curve attributable (every rank contributes, rank 0 prints): each rank's own time per
    step, the duration of its per-step collective, what it found, whether it was pinned.  Works on RCCL and gloo ranks."""
    import torch
    mine = torch.tensor([1e3 * dt_local / max(steps, 1), gather.gather_ms() if gather.gather_ms() is not None else -1.0,
                         float(found_local), 1.0 if pinned else 0.0, float(pinned["numa_node"]) if pinned else -1.0], dtype=torch.float64, device=dev)
    allr = torch.zeros((world, mine.numel()), dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_gather_into_tensor(allr.view(-1), mine)
    else:
        allr[0] = mine
    a = allr.cpu().tolist()
    return {"per_rank_ms_per_step": [r[0] for r in a], "gather_ms_per_step": [(r[1] if r[1] >= 0 else None) for r in a],
            "found_per_rank": [int(r[2]) for r in a], "ranks_pinned_to_numa": [bool(r[3]) for r in a], "numa_node_per_rank": [int(r[4]) for r in a],
            "collective_world": (dist.get_world_size() if dist is not None else 1),
            "collective_backend": (dist.get_backend() if dist is not None else "none")}


class _stdout_to_stderr:
    """RCCL prints a version banner on STDOUT when its first communicator is created; the contract of this script is ONE
    JSON line there.  While the process group comes up, file descriptor 1 points at stderr."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def launch_ranks(a):
    """--gpus N > 1 without a launcher: start N ranks as a child torchrun.  This parent never imports torch and never
    makes a HIP call -- the devices are counted from sysfs (a rank that finds no device of its own fails by itself)."""
    import socket
    n = visible_gpus_sysfs()
    if n < a.gpus:
        sys.stderr.write("bench.py: --gpus %d asked for, %d HIP device(s) visible: not running a smaller world in its place\n" % (a.gpus, n))
        return 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def workload_label(a, fid):
    what = "%dx%d square fiducials per frame" % fid if fid else "checkerboard"
    if fid:
        tag = "BASELINE.json configs[4]-style"
    elif a.fisheye:
        tag = "BASELINE.json configs[3]" if (a.width, a.height) == (3840, 2160) else "BASELINE.json configs[3]-style"
    elif (a.width, a.height, a.batch) == (1920, 1080, 1024):
        tag = "BASELINE.json configs[1]"
    else:
        tag = "configs[1]-style at another size"
    return "batch of %d synthetic %dx%d BGR8 %s frames per GPU, %s distortion, device-resident (%s); corners + PnP" % (
        a.batch, a.width, a.height, what, "fisheye" if a.fisheye else "plumb-bob", tag)


def run_steps(det, frames, B, gather, steps, sync_steps, want_corners=False):
    """The timed region's body: K steps of the whole path + the exchange of the records.  Works with any detector that
    has detect / submit / collect (tests/test_dist_gloo.py drives it with a stand-in on CPU ranks).
    Leaves run_steps.trace: per step the wall-clock interval between consecutive results reaching the host
    (`wall_ms`: step k = end of collect k-1 .. end of collect k; they add up to the region), and -- streaming form, from
    the detector's HIP events -- the device time of the batch (`device_ms`), the device's idle time in front of it
    (`idle_ms`: > 0 where the host submitted late) and the five stage times as they ran inside the step (`stages`)."""
    found = 0
    run_steps.local_found = 0        # records THIS rank produced in its last step (found: records visible after the exchange)
    tr = run_steps.trace = {"wall_ms": [], "device_ms": [], "idle_ms": [], "stages": []}
    # ranks that exchange device-packed tables keep the detector on the stream the collective is ordered with, and count
    # the gathered records (a host synchronisation) only after the last step
    kw = {"stream": gather.stream} if getattr(gather, "stream", None) is not None else {}
    t_prev = time.perf_counter()
    if sync_steps:
        for k in range(steps):
            getattr(gather, "before_submit", lambda s: None)(0)
            dets, _ = det.detect(frames, B, want_corners=want_corners, **kw)
            run_steps.local_found = len(dets)
            found = gather.exchange(dets, 0, k == steps - 1)
            t_now = time.perf_counter(); tr["wall_ms"].append(1e3 * (t_now - t_prev)); t_prev = t_now
            if hasattr(det, "last_timings"):
                tr["stages"].append(det.last_timings())
    else:
        # the streaming form of the same K steps (rcc_detect_batch_submit / _collect): batch k+1 is launched before
        # the host unpacks batch k, so the device does not idle during the unpack and the exchange of the records.
        # Every step's work -- all kernels, the device-to-host copy, the unpack, the all_gather -- is inside the region.
        slot_of_next = getattr(det, "_nsub", 0) & 1          # result slot (and record table) of the next submission
        before = getattr(gather, "before_submit", lambda s: None)
        # GPU ranks whose tables the detector packs: the collective is queued right behind the batch on the same stream
        inline = bool(getattr(gather, "inline", False) and getattr(gather, "attached", False) and gather.dist is not None)
        skw = dict(kw, want_corners=True) if want_corners else kw

        def sub(slot):
            before(slot); det.submit(frames, B, **skw)
            if inline:
                gather.exchange(None, slot, False)
        if steps > 0:
            sub(slot_of_next); slot_of_next ^= 1
        for k in range(steps):
            if k + 1 < steps:
                sub(slot_of_next); slot_of_next ^= 1
            dets, _ = det.collect()
            run_steps.local_found = len(dets)
            if inline:
                found = gather.count() if k == steps - 1 else -1
            else:
                found = gather.exchange(dets, getattr(det, "last_slot", 0), k == steps - 1)
            t_now = time.perf_counter(); tr["wall_ms"].append(1e3 * (t_now - t_prev)); t_prev = t_now
            if hasattr(det, "last_step_times"):
                st = det.last_step_times()
                tr["device_ms"].append(st.pop("device")); tr["idle_ms"].append(st.pop("idle_before")); tr["stages"].append(st)
    return found


def trace_summary(tr, B, world=1):
    """per-step figures of one timed region for the JSON line (this rank's own steps)"""
    w = tr.get("wall_ms") or []
    out = {"step_ms_all": [round(v, 4) for v in w]}
    if w:
        med = statistics.median(w)
        out.update({"step_ms_median": med, "step_ms_min": min(w), "step_ms_max": max(w), "value_median": world * B / (med * 1e-3),
                    "what": "wall clock between consecutive results reaching the host on rank 0 (they add up to the timed region); value_median = frames per step / median step -- `value` itself stays total frames / total time"})
    if tr.get("device_ms"):
        d, g = tr["device_ms"], tr["idle_ms"]
        out["device_ms_all"] = [round(v, 4) for v in d]
        out["device_idle_before_ms_all"] = [round(v, 4) for v in g]
        out["device_ms_median"] = statistics.median(d)
        late = [v for v in g[1:] if v > 0.1]
        out["device_idle_total_ms"] = sum(v for v in g[1:] if v > 0)
        out["steps_submitted_late"] = len(late)
        out["device_what"] = "HIP events on the step's stream: device_ms = the stream reaches the batch -> its records are in pinned host memory; device_idle_before = end of the previous batch -> begin of this one: about 0.01 ms while the host keeps one batch ahead (plus the in-line all_gather where ranks exchange records), more where the host submitted late (> 0.1 ms is counted); entry 0 is the start of the region (barrier + synchronize lie in it) and is left out of the two sums"
    if tr.get("host_during_region"):
        out["host_during_region"] = tr["host_during_region"]
        out["host_during_region_what"] = "rank 0's submitting thread over the K steps: context switches, the cgroup's quota-throttling counters (deltas), the CPU it ran on at both ends, and what Python's cyclic garbage collector did (every collection inside the region: generation, ms)"
    st = [x for x in (tr.get("stages") or []) if x and min(x.values()) >= 0]
    if st:
        out["stage_ms_mean"] = {k: sum(x[k] for x in st) / len(st) for k in st[0]}
        out["stage_ms_median"] = {k: statistics.median(x[k] for x in st) for k in st[0]}
        out["stage_ms_max"] = {k: max(x[k] for x in st) for k in st[0]}
        allst = tr.get("stages") or []
        if w and len(allst) == len(w) and allst[w.index(max(w))]:
            # which stage a slow step lost its time in (one step of ~200 on this pool shows the one-wave-per-frame lattice + pose
            # kernel taking 1.4 ms instead of 0.17: a device-side hiccup, no host cause in the counters)
            out["slowest_step"] = {"index": w.index(max(w)), "wall_ms": max(w), "stages": allst[w.index(max(w))]}
    return out


def host_cpus():
    """logical CPUs this process may use: affinity mask, capped by a cgroup quota if there is one"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(int(q) / int(p)))
    except Exception:
        pass
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    return n, quota, model


def _frame_cmp(cfg, fid, f, g_dets, g_fc, n, od, ofc, acc):
    """one frame, GPU records against the oracle's: index / id / status mismatches are counted, float differences maxed"""
    import numpy as np
    if fid:
        if n != len(g_dets):
            acc["mism"] += 1
            return
        for q in range(n):
            if g_dets[q].id != od[q].id:
                acc["mism"] += 1
                continue
            acc["corner"] = max(acc["corner"], float(np.abs(np.array(g_dets[q].corners) - np.array([[od[q].corners[c][0], od[q].corners[c][1]] for c in range(4)])).max()))
            acc["rvec"] = max(acc["rvec"], float(np.abs(np.array(list(g_dets[q].rvec)) - np.array(list(od[q].rvec))).max()))
            acc["tvec"] = max(acc["tvec"], float(np.abs(np.array(list(g_dets[q].tvec)) - np.array(list(od[q].tvec))).max()))
        return
    nc = cfg.board_cols * cfg.board_rows
    if (ofc.ncorners != g_fc.ncorners) or (ofc.status != g_fc.status) or (ofc.ncand != g_fc.ncand) or (ofc.nkept != g_fc.nkept) or (n != len(g_dets)):
        acc["mism"] += 1
        return
    if n:
        gp = np.array([[g_fc.px[k][0], g_fc.px[k][1]] for k in range(nc)])
        op = np.array([[ofc.px[k][0], ofc.px[k][1]] for k in range(nc)])
        acc["mism"] += int((gp != op).any())
        gx = np.array([[g_fc.xy[k][0], g_fc.xy[k][1]] for k in range(nc)])
        ox = np.array([[ofc.xy[k][0], ofc.xy[k][1]] for k in range(nc)])
        acc["corner"] = max(acc["corner"], float(np.abs(gx - ox).max()))
        acc["rvec"] = max(acc["rvec"], float(np.abs(np.array(list(g_dets[0].rvec)) - np.array(list(od.rvec))).max()))
        acc["tvec"] = max(acc["tvec"], float(np.abs(np.array(list(g_dets[0].tvec)) - np.array(list(od.tvec))).max()))


def cpu_and_accuracy_legs(a, out, det, cfg, frames, poses, B, fid, tpf):
    """rank 0, N = 1, after the timed region.  (1) cpu_baseline: the oracle (kind "port") timed on the host's cores on a
    bounded sample.  (2) accuracy_vs_oracle: every stage-level figure on the first frames AND on every frame the detector
    did not answer completely.  (3) a status pass of the oracle over ALL frames of the batch: a frame the detector rejects
    must be one the oracle rejects too -- the consumer silently skips an empty array (corner_detections.cpp:43-56), so a
    wrongly empty frame would be lost without a trace.  (4) the reference_mode variant (corners truncated to int before
    the pose solve, corner_detections.cpp:53-54)."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle import orc_py
    from robot_camera_calibration_amd import abi, api, synth
    ncpu, quota, model = host_cpus()
    T = max(1, min(ncpu, quota or ncpu, B))
    S = min(B, a.cpu_sample if a.cpu_sample > 0 else (4 * T if fid else 8 * T))
    SA = min(32, S)                      # frames compared in full whatever their status
    host = frames[:S].cpu().numpy()
    g_dets, g_fcs = det.detect(frames, B, want_corners=True)        # the GPU's answer for the WHOLE batch
    by = {}
    for d in g_dets:
        by.setdefault(int(d.frame), []).append(d)
    build = "-O3 -march=native"
    try:
        Lt = orc_py.native_library()
    except Exception:
        Lt, build = None, "-O2 (the -O3 -march=native build failed on this host)"
    ctxs = [orc_py.Context(cfg, Lt) for _ in range(T)]
    parts = [list(range(t, S, T)) for t in range(T)]

    def work(t):
        for f in parts[t]:
            ctxs[t].detect(host[f], f)
    rates = []
    with ThreadPoolExecutor(T) as ex:
        for _ in range(3):
            t1 = time.perf_counter()
            list(ex.map(work, range(T)))
            rates.append(S / (time.perf_counter() - t1))
    S1 = min(4, S)
    r1 = []
    for _ in range(3):
        t2 = time.perf_counter()
        for f in range(S1):
            ctxs[0].detect(host[f], f)
        r1.append(S1 / (time.perf_counter() - t2))
    out["cpu_baseline"] = {"value": statistics.median(rates), "unit": "frames/s", "cores": T, "single_thread_value": statistics.median(r1), "kind": "port",
                           "repetitions": 3, "all_rates": rates,
                           "sample": "%d of the same %dx%d frames through oracle/ (C, %s, %d threads over frames, median of 3 repetitions); host: %s, %d logical CPUs usable%s" % (
                               S, a.width, a.height, build, T, model or "unknown CPU", ncpu, (", cgroup quota %d" % quota) if quota else "")}

    # ---- (3) status pass over every frame of the batch (the timing build: same sources, same -ffp-contract=off)
    ostat = np.zeros((B, 3), np.int64)          # oracle: records, status, ncorners (board) / records, 0, sum of ids (tags)
    t3 = time.perf_counter()
    CH = 128
    for c0 in range(0, B, CH):
        hc = frames[c0:min(B, c0 + CH)].cpu().numpy()

        def swork(t, hc=hc, c0=c0):
            for i in range(t, len(hc), T):
                n, od, ofc = ctxs[t].detect(hc[i], c0 + i)
                ostat[c0 + i] = (n, ofc.status, ofc.ncorners) if not fid else (n, 0, sum(int(od[q].id) for q in range(n)))
        with ThreadPoolExecutor(T) as ex:
            list(ex.map(swork, range(T)))
        del hc
    full_rate = B / (time.perf_counter() - t3)
    gstat = np.zeros((B, 3), np.int64)
    for f in range(B):
        g = by.get(f, [])
        gstat[f] = (len(g), g_fcs[f].status, g_fcs[f].ncorners) if not fid else (len(g), 0, sum(int(d.id) for d in g))
    incomplete = [f for f in range(B) if (gstat[f, 0] < tpf)]                  # frames that did not yield every target
    status_mism = [f for f in range(B) if tuple(gstat[f]) != tuple(ostat[f])]
    out["cpu_baseline"]["full_batch"] = {"value": full_rate, "unit": "frames/s", "frames": B,
                                         "what": "the status pass over all %d frames (same build, %d threads, one pass, host copies of 128-frame chunks included)" % (B, T)}

    # ---- (2) full comparison: the first SA frames + every frame that is incomplete or whose status disagrees
    chk = orc_py.Context(cfg)                   # the checking build (oracle/liborc.so)
    extra = sorted(set(incomplete + status_mism) - set(range(SA)))[:96]
    acc = {"mism": 0, "corner": 0.0, "rvec": 0.0, "tvec": 0.0}
    gtc = gtr = gtt = 0.0
    Kb = np.array(list(cfg.K)); objb = synth.board_object_points(cfg.board_cols, cfg.board_rows, cfg.board_square)
    nc = cfg.board_cols * cfg.board_rows
    for f in list(range(SA)) + extra:
        hf = host[f] if f < S else frames[f].cpu().numpy()
        n, od, ofc = chk.detect(hf, f)
        _frame_cmp(cfg, fid, f, by.get(f, []), g_fcs[f], n, od, ofc, acc)
        if not fid and n and by.get(f):
            # informational: against the analytic ground truth of the synthetic camera (undistorted image =
            # pinhole projection of the board; the 9x7-square board has a 180-degree ambiguity)
            gx = np.array([[g_fcs[f].xy[k][0], g_fcs[f].xy[k][1]] for k in range(nc)])
            gt = synth.project_points(objb, poses[f][:3], poses[f][3:], Kb)
            flip = np.abs(gx - gt).max() > np.abs(gx - gt[::-1]).max()
            gtc = max(gtc, float(np.abs(gx - (gt[::-1] if flip else gt)).max()))
            Rg = synth.rodrigues(poses[f][:3]) @ (np.diag([-1.0, -1.0, 1.0]) if flip else np.eye(3))
            gtr = max(gtr, float(np.abs(synth.rodrigues(list(by[f][0].rvec)) - Rg).max()))
            gtt = max(gtt, float(np.abs(np.array(list(by[f][0].tvec)) - poses[f][3:]).max()))
    out["accuracy_vs_oracle"] = {"frames": SA + len(extra), "frames_what": "the first %d frames + %d incomplete / disagreeing frames of the batch" % (SA, len(extra)),
                                 "max_corner_err_px": acc["corner"], "max_rvec_err": acc["rvec"], "max_tvec_err": acc["tvec"],
                                 "corner_index_or_status_mismatches": acc["mism"],
                                 "not_found_frames": incomplete[:64], "not_found_count": len(incomplete),
                                 "status_mismatches_all_frames": len(status_mism), "status_mismatch_frames": status_mism[:64],
                                 "all_frames_what": "oracle status pass over all %d frames: records per frame, frame status and corner count%s equal the detector's" % (B, " (tags: count and id sum)" if fid else "")}
    if fid:
        # informational: against the rendered tags (the grid's layout through the frame's pose), first SA frames
        (fhx, fhy), centres, ids = synth.fiducial_grid_layout(fid[0], fid[1], cfg.tag_size)
        objt = synth.tag_object_points(cfg.tag_size)
        ce, te = [], []
        for f in range(SA):
            Rg = synth.rodrigues(poses[f][:3])
            for d in by.get(f, []):
                c = centres[list(ids).index(int(d.id))]
                gt = synth.project_points(objt + c, poses[f][:3], poses[f][3:], Kb)
                ce.append(np.abs(np.array(d.corners) - gt).max())
                te.append(np.abs(np.array(list(d.tvec)) - (Rg @ c + poses[f][3:])).max())
        if ce:
            out["accuracy_vs_ground_truth"] = {"frames": SA, "tags": len(ce), "max_corner_err_px": float(max(ce)), "median_corner_err_px": float(np.median(ce)),
                                               "p90_corner_err_px": float(np.percentile(ce, 90)), "max_tvec_err_m": float(max(te)), "median_tvec_err_m": float(np.median(te)),
                                               "note": "informational: detector error on noisy supersampled renders (per tag: largest coordinate error of its four corners), not a parity figure"}
    if not fid:
        out["accuracy_vs_ground_truth"] = {"frames": SA + len(extra), "max_corner_err_px": gtc, "max_rotation_matrix_err": gtr, "max_tvec_err_m": gtt,
                                           "note": "informational: detector error on noisy supersampled renders, not a parity figure"}

    # ---- (4) reference_mode: the reference casts the corners to int before solvePnP (corner_detections.cpp:53-54).
    # Same frames, a detector and an oracle context with cfg.reference_mode = 1: parity of that variant, and what the
    # truncation costs against the rendered pose (informational).
    try:
        cfgr = api.clone_config(cfg)
        cfgr.reference_mode = 1
        cfgr.batch_capacity = SA
        detr = api.Detector(cfgr)
        rd, rfc = detr.detect(frames[:SA].contiguous(), SA, want_corners=True)
        detr.close()
        rby = {}
        for d in rd:
            rby.setdefault(int(d.frame), []).append(d)
        ocfg = api.clone_config(cfgr)
        chkr = orc_py.Context(ocfg)
        racc = {"mism": 0, "corner": 0.0, "rvec": 0.0, "tvec": 0.0}
        dr = dt_ = 0.0
        for f in range(SA):
            n, od, ofc = chkr.detect(host[f], f)
            _frame_cmp(cfgr, fid, f, rby.get(f, []), rfc[f], n, od, ofc, racc)
            for q, d in enumerate(rby.get(f, [])):              # against the sub-pixel pose of the same target
                sub = [x for x in by.get(f, []) if x.id == d.id]
                if sub:
                    dr = max(dr, float(np.abs(synth.rodrigues(list(d.rvec)) - synth.rodrigues(list(sub[0].rvec))).max()))
                    dt_ = max(dt_, float(np.abs(np.array(list(d.tvec)) - np.array(list(sub[0].tvec))).max()))
        chkr.close()
        out["accuracy_reference_mode"] = {"frames": SA, "what": "cfg.reference_mode = 1: corners truncated to int before the pose solve, as corner_detections.cpp:53-54 does",
                                          "max_rvec_err_vs_oracle": racc["rvec"], "max_tvec_err_vs_oracle": racc["tvec"], "mismatches": racc["mism"],
                                          "max_rotation_matrix_shift_vs_subpixel_pose": dr, "max_tvec_shift_vs_subpixel_pose_m": dt_}
    except Exception as e:
        out["accuracy_reference_mode"] = {"error": repr(e)}
    chk.close()
    for c in ctxs:
        c.close()


def main():
    a = parse()
    if a.gpus < 1:
        sys.stderr.write("bench.py: --gpus must be >= 1\n")
        return 2
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        return launch_ranks(a)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d\n" % (a.gpus, world))
        return 2

    # thread pools (OpenMP / BLAS behind numpy and torch) size themselves by the CPUs they SEE; under a cgroup quota that is far
    # more than the process may use, and a pool of spinning workers can eat the quota and get the whole process throttled
    # for the rest of the scheduler period -- a multi-millisecond stall of the submitting thread.  Size them by what is usable.
    ncpu_, quota_, model_ = host_cpus()
    usable = max(1, min(ncpu_, quota_ or ncpu_))
    pools = {}
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        pools[var] = os.environ.setdefault(var, str(usable))
    host_info = {"cpu": model_, "cpus_visible": ncpu_, "cgroup_quota_cpus": quota_, "thread_pools": pools}

    import numpy as np
    import torch
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    if local >= torch.cuda.device_count():
        sys.stderr.write("bench.py: rank %d wants device %d, %d visible\n" % (rank, local, torch.cuda.device_count()))
        return 2
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    pinned = pin_to_gpu_numa(torch, local) if not a.no_pin else None
    host_info["pinned"] = pinned
    dist = None
    if "WORLD_SIZE" in os.environ:          # launched by torchrun: the distributed path, whatever the world size (a world of
        import torch.distributed as dist    # one still runs the RCCL calls: init, barrier, all_gather of the record tables)
        with _stdout_to_stderr():
            dist.init_process_group(backend="nccl", device_id=dev)   # "nccl" is RCCL on ROCm
            dist.barrier()                                           # the communicator (and RCCL's banner) comes up here
            torch.cuda.synchronize()
        if dist.get_world_size() != a.gpus:
            sys.stderr.write("bench.py: RCCL world of %d ranks, --gpus %d\n" % (dist.get_world_size(), a.gpus))
            return 2
        world = dist.get_world_size()

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
        cfg.tag_refine = abi.RCC_TAG_REFINE_EDGES if a.tag_refine == "edges" else abi.RCC_TAG_REFINE_CORNER_SUBPIX
    if a.min_contrast >= 0:
        cfg.thr_min_contrast = a.min_contrast
    if a.harris_thresh >= 0:
        cfg.harris_thresh = a.harris_thresh
    cfg.device = local
    cfg.batch_capacity = a.batch
    det = api.Detector(cfg)
    det.set_pipeline(a.pipeline)
    det.set_dense_variant(a.dense_variant)
    det.set_ingest_variant(a.ingest_variant)
    if a.tail_overlap:
        det.set_tail_overlap(a.tail_overlap)
    B = a.batch
    px = a.width * a.height

    # ---- workload: B distinct frames per rank, rendered on the device (not timed)
    sp = abi.default_synth_params() if a.noise < 0 else abi.default_synth_params(noise=a.noise)
    if a.blur or a.shade:
        sh = [int(v) for v in a.shade.split(",")] if a.shade else [0, 0, 0]
        abi.set_optics(sp, ("3tap" if a.blur == "3tap" else float(a.blur)) if a.blur else None, *sh)
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

    tpf = fid[0] * fid[1] if fid else 1
    gather = rdist.PoseGather(B, dev, world, dist, rank, targets_per_frame=tpf, inline=not a.gather_side_stream)
    if dist is not None:
        gather.attach(det, frame_offset=first)      # the detector packs the records on the device, every batch
        # torch loads a kernel's code object the first time it is launched: the compare + sum behind gather.count() took ~65 ms when
        # they first ran at the end of the warm-up -- the device idled that long in front of the timed region, its clock fell, and the
        # region's first four steps ran at 3.2 / 3.2 / 3.0 / 2.9 ms instead of 2.75 (steps_trace.device_idle_before_ms_all[0] of the
        # world-of-one runs of rounds 3-4).  Run them once here, long before the warm-up.
        gather.count()
        torch.cuda.synchronize()

    def timed_region(sync_form, want_corners=False):
        """W warm-up steps IN THE FORM THAT IS TIMED (so that the code paths, pinned slots and event rings of that form have all
        been through once), barrier + synchronize, exactly K steps, synchronize + barrier; max over ranks.  Nothing that takes
        time may stand between the warm-up and the region: a full garbage collection there (35 ms with torch imported) left the device
        idle for 39 ms, its clock fell, and the region's first eight steps ran at 3.14, 3.21, 2.97, 2.81 ... ms of DEVICE time instead
        of 2.65 (`steps_trace.device_idle_before_ms_all[0]` shows the gap).  The collection therefore runs BEFORE the warm-up."""
        import gc
        gc.collect()
        hc0 = host_counters()
        run_steps(det, frames, B, gather, a.warmup, sync_form, want_corners)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        gather.reset_timing()
        with gc_watch(a.keep_gc):
            t0 = time.perf_counter()
            fnd = run_steps(det, frames, B, gather, a.steps, sync_form, want_corners)
            tr = run_steps.trace
            fl = run_steps.local_found
            torch.cuda.synchronize()
            dl = time.perf_counter() - t0          # this rank's own K steps, before it waits for the others
            gcs = list(gc_watch.events)
        hc1 = host_counters()
        tr["host_during_region"] = {k: (hc1[k] - hc0[k]) if k != "cpu" else [hc0[k], hc1[k]] for k in hc1 if k in hc0}
        tr["host_during_region"]["python_gc"] = "running" if a.keep_gc else "collected before the warm-up, off for the region"
        tr["host_during_region"]["counters_span"] = "warm-up + region"
        tr["host_during_region"]["python_gc_collections_generation_ms"] = gcs
        if dist is not None:
            dist.barrier()
        d = time.perf_counter() - t0
        rep = rank_report(dist, dev, world, dl, a.steps, fl, gather, pinned)
        tmax = torch.tensor([d], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return {"dt": float(tmax.item()), "found": fnd, "found_local": fl, "trace": tr, "ranks": rep}

    # the region that defines `value` (streamed unless --sync-steps) and the other form for comparison; --sync-first swaps
    # the order of the two (the first region after start-up meets whatever the device and the host have not ramped up yet)
    regions = {}
    order = [a.sync_steps] if a.sync_steps else ([True, False] if a.sync_first else [False, True])
    for form in order:
        regions[form] = timed_region(form)
    main_r = regions[a.sync_steps]
    dt, found, found_local, ranks = main_r["dt"], main_r["found"], main_r["found_local"], main_r["ranks"]
    dense_in_step = [x["dense"] for x in main_r["trace"]["stages"] if x and x.get("dense", -1) > 0]
    sync_fps = (world * B * a.steps / regions[True]["dt"]) if (True in regions and not a.sync_steps) else None
    # the same streamed steps with the per-frame corner tables (rcc_frame_corners, 6 160 bytes per frame) copied back too
    corner_r = timed_region(False, want_corners=True) if (not a.sync_steps and not a.no_extra_legs) else None
    # the engine clock the chip holds under a vector-issue load, and what a vector wave-instruction costs, measured right
    # behind the timed regions (rank 0)
    clock = None
    if rank == 0:
        try:
            clock = det.measure_clock(6, 20.0)
            clock["sysfs"] = sysfs_clocks(torch, local)
        except Exception as e:
            clock = {"error": repr(e)}
    # per-stage times: one extra (untimed) step as a single pass on one stream -- in the pipelined step the stages of
    # different chunks overlap, so they have no separate durations
    prev = det.set_pipeline(1)
    run_steps(det, frames, B, gather, 1, True)
    timings = det.last_timings()
    step_dense_kernel = det.last_dense_kernel()
    det.set_pipeline(prev)

    out = None
    if rank == 0:
        fps = world * B * a.steps / dt
        out = {
            "metric": "calibration frames/sec at %dx%d" % (a.width, a.height), "value": fps, "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/i32 pixel stages, f64 sub-pixel + PnP", "data": "synthetic",
            "config": {"workload": workload_label(a, fid),
                       "frames_per_step_per_gpu": B,
                       "target": ("%dx%d square fiducials of 0.10 m per frame (build family36b), corners by %s, 4-point PnP per tag" % (fid + ("refine_edges" if a.tag_refine == "edges" else "cornerSubPix",))) if fid else "8x6 inner-corner checkerboard, 0.108 m", "distortion": ("fisheye" if a.fisheye else "plumb-bob") + ", undistort on",
                       "detector": {"thr_min_contrast": int(cfg.thr_min_contrast), "harris_thresh": int(cfg.harris_thresh)},
                       "camera_optics": {"noise_sigma": float(sp.noise_sigma), "blur_taps": list(sp.blur_taps), "shade_x_permille": int(sp.shade_x_permille), "shade_y_permille": int(sp.shade_y_permille), "vignette_permille": int(sp.vignette_permille)},
                       "parallelism": "frame-sharded, 1 process per GPU, 1 all_gather of pose records (19 doubles per target slot, packed on the device) per step"},
            "targets_found_in_last_step": int(found), "targets_expected_per_step": int(world * B * tpf), "stage_ms_single_pass": timings, "pipeline_chunks": a.pipeline, "step_form": "sync detect()" if a.sync_steps else "submit/collect, one batch ahead", "value_with_sync_steps": sync_fps,
            "region_order": ["sync detect()" if f else "submit/collect" for f in order], "warmup_form": "the form that is timed (W steps in front of each region)",
            "steps_trace": trace_summary(main_r["trace"], B, world),
            "ranks": ranks, "host": host_info,
        }
        if True in regions and not a.sync_steps:
            out["sync_steps_trace"] = trace_summary(regions[True]["trace"], B, world)
        if corner_r is not None:
            out["value_with_corner_tables"] = {"value": world * B * a.steps / corner_r["dt"], "unit": "frames/s", "ms_per_step": 1e3 * corner_r["dt"] / a.steps,
                                               "what": "the same K streamed steps with want_corners: every frame's rcc_frame_corners table (status, counts, 48 integer + 48 sub-pixel corners; %.1f MB per batch) copied to the host beside the records" % (B * 6160 / 1e6),
                                               "steps_trace": trace_summary(corner_r["trace"], B, world)}
        out["clock"] = clock

    # ---- roofline of the threshold+corner pass (the pass BASELINE.json's north_star names) AS THE STEP RUNS IT, and of
    # the ingest pass: algorithmic bytes / HIP-event time on the launch stream.  rcc_detect_batch leaves the binary image
    # as a one-byte-per-4x4-tile threshold map, so the kernel of the step reads px and writes px/16 bytes per frame;
    # `frac` is quoted on those (its own) bytes, `frac_2px` on the 2*px of SURVEY.md 8(d) (read grey + write the binary
    # image), and `stage_form` is the same pass writing the full image (rcc_stage_threshold_corner: another kernel).
    if rank == 0:
        grey = torch.empty((B, px), dtype=torch.uint8, device=dev)
        binm = torch.empty((B, px), dtype=torch.uint8, device=dev)
        cand = torch.empty((B, cfg.max_candidates * 8), dtype=torch.uint8, device=dev)
        cnt = torch.empty((B,), dtype=torch.int32, device=dev)
        det.stage_ingest(frames, B, grey)
        det.time_dense(grey, B, None, cand, cnt, 1)
        ms_b2b = det.time_dense(grey, B, None, cand, cnt, a.roofline_reps)
        k_step = det.last_dense_kernel()
        # the figure of record is the kernel's duration INSIDE the timed steps (HIP events around its launch on the step's
        # stream, mean over the K steps); launched back to back with itself it is ~0.1 ms shorter (nothing of the ingest
        # pass's 2 GB of stores still draining), reported beside it
        ms_c = (sum(dense_in_step) / len(dense_in_step)) if dense_in_step and min(dense_in_step) > 0 else ms_b2b
        det.time_dense(grey, B, binm, cand, cnt, 1)
        ms = det.time_dense(grey, B, binm, cand, cnt, a.roofline_reps)
        k_stage = det.last_dense_kernel()
        alg2 = 2.0 * px * B
        algc = (1.0 + 1.0 / 16.0) * px * B

        def traffic_of(name):
            tpath = os.path.join(ROOT, "profiles", name)
            if os.path.exists(tpath) and (a.width, a.height) == (1920, 1080) and not a.fisheye:
                try:
                    j = json.load(open(tpath))
                    return j.get("hbm_bytes_per_frame") * B, "profiles/%s <- profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of %s, taken on the %s (another box), scaled per frame; not measured in this run" % (name, j.get("source"), j.get("kernel"), j.get("code", "code of an earlier run"))
                except Exception:
                    pass
            return None, "none"
        props = torch.cuda.get_device_properties(dev)
        simds = int(props.multi_processor_count) * 4
        # cost of a vector wave-instruction per SIMD: measured in this process (out["clock"], k_probe.hip) -- the instruction
        # classes of the pass at 6 waves per SIMD, ns from HIP events, the clock from s_memtime / s_memrealtime inside that
        # kernel; the committed microbenchmark (4.4 cycles at the 2.4 GHz of the device properties) only if the probe failed
        if clock and "ns_per_wave_inst_per_simd" in clock:
            ns_inst, clock_mhz, cyc = clock["ns_per_wave_inst_per_simd"], clock["clock_mhz"], clock["cycles_per_wave_inst_per_simd"]
            cost_src = "measured in this run behind the timed regions (rcc_debug_measure_clock: packed 16-bit / dot2 / perm / DPP / add3 loop, 6 waves per SIMD, %.1f ms launch; clock = d s_memtime / d s_memrealtime x 100 MHz, median over the waves)" % clock["probe_ms"]
            clock_src = "measured in this run (in-kernel s_memtime / s_memrealtime of the issue probe)"
        else:
            clock_mhz = float(getattr(props, "clock_rate", 2400000)) / 1e3
            cyc = 4.4
            ns_inst = cyc / clock_mhz * 1e3
            cost_src = "profiles/r02_vbench.txt, r02_vbench_ilp.txt (another box) at the clock of the device properties: NOT measured in this run"
            clock_src = "device properties (nominal)"

        def issue_of(name, ms_launch):
            """the vector-issue roofline of the pass: SQ instruction counts of a committed profile x the measured issue cost"""
            ipath = os.path.join(ROOT, "profiles", name)
            if not (os.path.exists(ipath) and (a.width, a.height) == (1920, 1080) and not a.fisheye and not fid):
                return None
            try:
                j = json.load(open(ipath))
                valu, salu = float(j["valu_wave_insts_per_frame"]), float(j["salu_wave_insts_per_frame"])
                bound_ms = valu * B * ns_inst / simds * 1e-6
                return {"valu_wave_insts_per_frame": valu, "salu_wave_insts_per_frame": salu, "ns_per_inst": ns_inst, "cycles_per_inst": cyc,
                        "cycles_per_inst_source": cost_src,
                        "simds": simds, "clock_mhz": clock_mhz, "clock_source": clock_src, "issue_bound_ms": bound_ms, "frac_of_issue_bound": bound_ms / ms_launch,
                        "source": "profiles/%s <- profiles/%s: rocprofv3 --pmc SQ_INSTS_VALU / SQ_INSTS_SALU of %s, taken on the %s, scaled per frame; not measured in this run" % (name, j.get("source"), j.get("kernel"), j.get("code", "code of an earlier run"))}
            except Exception:
                return None
        tr_c, src_c = traffic_of("traffic_dense_step.json")
        tr_s, src_s = traffic_of("traffic_dense.json")
        # yardstick measured in the same process: a plain streaming copy of the same bytes (grey -> binary buffer)
        copy_ms = det.time_copy(grey, binm, B * px, a.roofline_reps) if (B * px) % 16 == 0 else None
        iss_c, iss_s = issue_of("issue_dense_step.json", ms_c), issue_of("issue_dense.json", ms)
        hbm_frac_traffic = (tr_c / (ms_c * 1e-3) / 1e9 / HBM_PEAK_GBS) if tr_c else (algc / (ms_c * 1e-3) / 1e9 / HBM_PEAK_GBS)
        bound = "valu-issue" if (iss_c and iss_c["frac_of_issue_bound"] > hbm_frac_traffic) else "hbm"
        out["roofline"] = {
            "bound": bound, "bound_what": "the resource the kernel runs closest to: vector-instruction issue (roofline.issue) vs HBM (achieved / peak / frac below are the HBM figures on algorithmic bytes, as BASELINE.json's metric asks; frac_hbm_on_traffic counts the bytes the counters saw)",
            "frac_hbm_on_traffic": hbm_frac_traffic, "issue": iss_c, "kernel": k_step, "kernel_in_timed_step": step_dense_kernel,
            "what": "threshold+corner pass as rcc_detect_batch launches it (binary image kept as a 1-byte-per-4x4-tile threshold map)",
            "achieved": algc / (ms_c * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": algc / (ms_c * 1e-3) / 1e9 / HBM_PEAK_GBS, "frac_2px": alg2 / (ms_c * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "frac_of_guide_copy_6290": algc / (ms_c * 1e-3) / 1e9 / 6290.0,      # both peaks -- the specification's 8 TB/s (frac) and the guide's measured float4 copy
            "alg_bytes_per_launch": algc, "alg_bytes_2px_per_launch": alg2, "ms_per_launch": ms_c, "frames_per_launch": B,
            "ms_per_launch_source": ("HIP events around the launch inside each of the %d timed steps (mean)" % len(dense_in_step)) if dense_in_step and min(dense_in_step) > 0 else "back-to-back launches (no in-step events in this mode)",
            "ms_per_launch_back_to_back": ms_b2b, "frac_back_to_back": algc / (ms_b2b * 1e-3) / 1e9 / HBM_PEAK_GBS, "frac_2px_back_to_back": alg2 / (ms_b2b * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "traffic": tr_c, "traffic_source": src_c,
            "stage_form": {"kernel": k_stage, "issue": iss_s, "what": "the same pass writing the full binary image (rcc_stage_threshold_corner), 2*px algorithmic bytes per frame",
                           "ms_per_launch": ms, "alg_bytes_per_launch": alg2, "achieved": alg2 / (ms * 1e-3) / 1e9,
                           "frac": alg2 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "frac_of_guide_copy_6290": alg2 / (ms * 1e-3) / 1e9 / 6290.0,
                           "traffic": tr_s, "traffic_source": src_s,
                           "copy_same_bytes_ms": copy_ms, "copy_GBps": (alg2 / (copy_ms * 1e-3) / 1e9) if copy_ms else None,
                           "frac_of_copy": (copy_ms / ms) if copy_ms else None}}
        det.time_ingest(frames, B, grey, 1)
        msi = det.time_ingest(frames, B, grey, max(1, a.roofline_reps // 2))
        algi = 4.0 * px * B
        tr_i, src_i = traffic_of("traffic_ingest.json")
        out["roofline_ingest"] = {"bound": "hbm", "kernel": "k_ingest_staged<3> (undistort + grey)", "achieved": algi / (msi * 1e-3) / 1e9,
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": algi / (msi * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "frac_of_guide_copy_6290": algi / (msi * 1e-3) / 1e9 / 6290.0,
                                  "frac_of_own_copy": ((alg2 / (copy_ms * 1e-3)) and (algi / (msi * 1e-3)) / (alg2 / (copy_ms * 1e-3))) if copy_ms else None,
                                  "traffic": tr_i, "traffic_source": src_i, "alg_bytes_per_launch": algi, "ms_per_launch": msi}
        # the whole step against HBM: SURVEY 8(d)'s algorithmic bytes of the three passes over pixels (ingest 4 px, threshold + corner
        # pass px + px / 16 as the step runs it; the tail's bytes are negligible) over the step's time
        alg_step = (4.0 + 1.0 + 1.0 / 16.0) * px * B
        out["roofline_step"] = {"what": "all passes of one step: algorithmic bytes (ingest 4 px + threshold/corner pass (1 + 1/16) px per frame) / ms_per_step; the step also holds the issue-bound corner arithmetic and the latency-bound tail, so this is a lower bound on how well the two streaming passes use HBM",
                                "alg_bytes_per_step": alg_step, "achieved": alg_step / (1e-3 * out["ms_per_step"]) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": alg_step / (1e-3 * out["ms_per_step"]) / 1e9 / HBM_PEAK_GBS}
        del grey, binm, cand, cnt

    # ---- the two other rates SURVEY.md 8(d) asks for (rank 0, N=1): frames handed over in HOST memory (PCIe inside
    # the timed region; never `value`), and the latency of ONE frame through a batch-1 handle with host input -- the
    # cadence of the reference's consumer (corner_detections.cpp:41-65 takes one message at a time) and what
    # host/tag_detections_shim.cpp does per image.
    if rank == 0 and world == 1 and not a.no_extra_legs:
        try:
            hostf = torch.empty((B, cfg.frame_bytes), dtype=torch.uint8, pin_memory=True)
            hostf.copy_(frames)
            torch.cuda.synchronize()

            def timed(fn, reps=5):
                fn()
                ts = []
                for _ in range(reps):
                    t1 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t1)
                return statistics.median(ts), ts
            # what the link itself delivers for these bytes: the batch copied in 64-MiB pieces on two streams, nothing else
            dstb = torch.empty_like(frames)
            cs = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
            flat_h, flat_d = hostf.view(-1), dstb.view(-1)

            def copy_only():
                step = 64 << 20
                for k, o in enumerate(range(0, flat_h.numel(), step)):
                    with torch.cuda.stream(cs[k & 1]):
                        flat_d[o:o + step].copy_(flat_h[o:o + step], non_blocking=True)
                torch.cuda.synchronize()
            t_copy, _ = timed(copy_only)
            del dstb
            det.set_host_chunk(0)
            dth, all_p = timed(lambda: det.detect(hostf, B, want_corners=False))
            det.set_host_chunk(-1)
            dth1, _ = timed(lambda: det.detect(hostf, B, want_corners=False), 3)
            det.set_host_chunk(0)
            link = B * cfg.frame_bytes / t_copy / 1e9
            out["value_with_h2d"] = {"value": B / dth, "unit": "frames/s", "ms_per_step": 1e3 * dth, "repetitions": len(all_p), "all_ms": [1e3 * t for t in all_p],
                                     "what": "the same step with the batch in pinned HOST memory (RCC_MEM_HOST): chunks of ~192 MiB copied on two streams, each chunk's kernels under the next chunks' copies (median of 5)",
                                     "h2d_GBps": B * cfg.frame_bytes / dth / 1e9,
                                     "one_copy_then_kernels": {"value": B / dth1, "ms_per_step": 1e3 * dth1, "what": "rounds 1-2: one hipMemcpyAsync of the batch, then the kernels"},
                                     "link_measured_GBps": link, "link_measured_frames_per_s": link * 1e9 / cfg.frame_bytes,
                                     "link_measured_what": "the same bytes copied host -> device in 64-MiB pieces on two streams, no kernels (median of 5): what the PCIe path of this box delivers",
                                     "frac_of_measured_link": (B / dth) / (link * 1e9 / cfg.frame_bytes),
                                     "pcie_ceiling_frames_per_s": PCIE_GBS * 1e9 / cfg.frame_bytes, "pcie_GBps_spec": PCIE_GBS}
            # the same with MONO8 frames (1/3 of the bytes per frame): the grey planes of the batch as host input
            try:
                cfgm = api.clone_config(cfg)
                abi.set_geometry(cfgm, a.width, a.height, abi.RCC_PIX_MONO8)
                for i in range(9):
                    cfgm.K[i] = cfg.K[i]
                detm = api.Detector(cfgm)
                hostm = torch.empty((B, cfgm.frame_bytes), dtype=torch.uint8, pin_memory=True)
                hostm.copy_(frames.view(B, -1, 3)[:, :, 1])            # the green plane: a mono rendition of the same scenes
                torch.cuda.synchronize()
                dm, _ = detm.detect(hostm, B, want_corners=False)
                dtm, _ = timed(lambda: detm.detect(hostm, B, want_corners=False))
                out["value_with_h2d"]["mono8"] = {"value": B / dtm, "unit": "frames/s", "ms_per_step": 1e3 * dtm, "h2d_GBps": B * cfgm.frame_bytes / dtm / 1e9,
                                                  "targets_found": len(dm), "what": "the same scenes as MONO8 frames in pinned host memory (2.07 MB per frame)"}
                detm.close()
                del hostm
            except Exception as e:
                out["value_with_h2d"]["mono8"] = {"error": repr(e)}
            del hostf
            cfg1 = api.clone_config(cfg)
            cfg1.batch_capacity = 1
            det1 = api.Detector(cfg1)
            one = torch.empty((1, cfg.frame_bytes), dtype=torch.uint8, pin_memory=True)
            one.copy_(frames[:1])
            torch.cuda.synchronize()
            for _ in range(5):
                det1.detect(one, 1, want_corners=False)
            lat = []
            for _ in range(50):
                t1 = time.perf_counter()
                d1, _ = det1.detect(one, 1, want_corners=False)
                lat.append(1e3 * (time.perf_counter() - t1))
            det1.close()
            out["latency_ms_b1"] = {"median": statistics.median(lat), "min": min(lat), "max": max(lat), "reps": len(lat), "targets_found": len(d1),
                                    "what": "one %dx%d frame in pinned host memory through a batch_capacity = 1 handle, call to return (H2D, all kernels, D2H)" % (a.width, a.height)}
        except Exception as e:      # a leg that cannot run (e.g. pinned allocation refused) must not lose the line
            out["extra_legs_error"] = repr(e)

    # ---- CPU baseline (the oracle = "port"; the reference's OpenCV path cannot be built here) and
    # accuracy against it, rank 0 at N=1 only
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu_and_accuracy_legs(a, out, det, cfg, frames, poses, B, fid, tpf)

    if rank == 0:
        print(json.dumps(out))
    det.close()
    if dist is not None:
        dist.barrier()          # rank 0's roofline legs run after the timed region: every rank leaves together
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
