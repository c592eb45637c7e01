"""Multi-GPU: frame sharding and the one collective of the path.

Frames are independent units (the tag-map accumulation of real_preprocessing/src/camera_pose.cpp:176-203
is downstream of this path), so N ranks simply own disjoint frame ranges and no collective touches
pixels.  The only exchange is the per-batch all-gather of fixed-size pose records so that one
process (the ROS node's publisher, or the map builder) sees every frame's result -- the role the
"tag_detections" topic plays between processes in the reference (corner_detections.cpp:78).
Records are REC = 19 doubles per slot (~156 KB per rank for 1024 frames): latency-bound on xGMI,
so it is ONE all_gather per batch, never per frame.

Record layout (include/rcc.h, RCC_REC_DOUBLES): valid, global frame, id, ncorners, rvec[3], tvec[3], rms and ALL
FOUR corners bl, br, tr, tl (x, y) -- the consumer reads all four (corner_detections.cpp:51-56).
On a GPU the table is packed on the device by the detector itself (rcc_set_record_tables, csrc/k_records.hip) and
handed to the collective as it is; `pack` is the host form of the same layout (CPU ranks of the gloo tests, and the
check of the device form in the GPU tests).  librcc_dist.so (include/rcc_dist.h) is the same all-gather for C++ hosts.
"""
import numpy as np
import torch

from . import abi

REC = abi.RCC_REC_DOUBLES  # 19


def shard_range(nframes, rank, world):
    """Contiguous block partition of frame indices: rank r owns [lo, hi)."""
    base, rem = divmod(nframes, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def pack(dets, nframes, targets_per_frame=1, frame_offset=0):
    """rcc_detection records (ordered by frame, as detect() returns them) -> (nframes * targets_per_frame, REC)
    float64, zero-padded; slot = frame * targets_per_frame + q for the frame's q-th record.  A frame with more
    records than targets_per_frame, or a frame index outside the batch, is an error -- never a silent cut."""
    nslots = nframes * targets_per_frame
    a = np.zeros((nslots, REC), np.float64)
    if len(dets) == 0:
        return a
    d = np.asarray(dets)
    fr = np.asarray(d["frame"], np.int64)
    if fr.min() < 0 or fr.max() >= nframes:
        raise ValueError("record of frame %d outside the batch of %d frames" % (int(fr.max() if fr.max() >= nframes else fr.min()), nframes))
    if np.any(np.diff(fr) < 0):
        raise ValueError("records are not ordered by frame")
    first = np.searchsorted(fr, fr, side="left")           # index of the frame's first record
    q = np.arange(len(d)) - first
    if q.max() >= targets_per_frame:
        raise ValueError("a frame holds %d records but the table has %d slots per frame" % (int(q.max()) + 1, targets_per_frame))
    slot = fr * targets_per_frame + q
    a[slot, 0] = 1.0
    a[slot, 1] = fr + frame_offset
    a[slot, 2] = d["id"]
    a[slot, 3] = d["ncorners"]
    a[slot, 4:7] = d["rvec"]
    a[slot, 7:10] = d["tvec"]
    a[slot, 10] = d["rms"]
    a[slot, 11:19] = np.asarray(d["corners"]).reshape(len(d), 8)
    return a


class PoseGather:
    """One all_gather of pose records per batch.  Without a process group (dist_module None: a single plain process) there
    is no collective at all; with one, the collective runs whatever the world size."""

    def __init__(self, nframes, device, world, dist_module=None, rank=0, targets_per_frame=1, inline=True):
        self.nframes, self.tpf = nframes, targets_per_frame
        self.nslots = nframes * targets_per_frame
        self.device, self.world, self.dist, self.rank = device, world, dist_module, rank
        on_gpu = getattr(device, "type", "cpu") == "cuda"
        # two send tables: the detector fills one per result slot (submit/collect keeps two batches in flight)
        self.tables = [torch.zeros((self.nslots, REC), dtype=torch.float64, device=device) for _ in range(2 if on_gpu else 1)]
        # one receive buffer per send table: the records of the batch in result slot i stay readable (gathered(i)) until the next
        # batch of that slot is exchanged -- with one batch ahead, batch k's records survive the queued exchange of batch k + 1
        self.recvs = [torch.zeros((world * self.nslots, REC), dtype=torch.float64, device=device) for _ in self.tables] if dist_module is not None else None
        self.recv = self.recvs[0] if self.recvs else None           # the buffer of the LAST exchange
        self.attached = False
        self.detector = None
        self.frame_offset = 0
        # GPU ranks, inline (default): the collective is queued on the detector's stream right behind the batch that packed the
        # table -- it runs in the gap between that batch's tail and the next batch's head (tens of microseconds for 156 KB).
        # inline = False: on a side stream, so that the detector's stream never waits for it; measured with a world of one,
        # the collective's kernel then waits ~0.6 ms for slots beside the next batch's ingest pass (the dispatcher serves the
        # big grid first) and the step is 0.11 ms longer.  `done[i]` marks the end of the last gather that read table i.
        self.inline = bool(inline) and on_gpu
        self.side = torch.cuda.Stream(device) if (on_gpu and not self.inline) else None
        self.done = [None, None]
        # duration of every collective since reset_timing(): event pairs on the side stream (GPU ranks), wall clock (CPU ranks)
        self._gather_ev, self._gather_s = [], []

    def attach(self, detector, frame_offset=0):
        """GPU ranks: let the detector pack its records into this gather's tables on the device"""
        detector.set_record_tables(self.tables[0], self.tables[-1], frame_offset)
        self.attached, self.frame_offset, self.detector = True, frame_offset, detector

    def exchange(self, dets, slot=0, count=True):
        """One step's exchange, whatever the rank is made of: the device table of result slot `slot` when a detector
        packs it (attach), else the host records; no collective at all without a process group.  Returns the number of
        valid records this rank sees -- or -1 with count=False on a GPU rank, which then only ENQUEUES the collective
        (no host synchronisation: the caller keeps the detector on the stream the collective is ordered with,
        `stream`, so nothing overwrites a table before it has been gathered)."""
        if self.dist is None:
            return len(dets)
        if self.attached:
            t = self.tables[slot if slot < len(self.tables) else 0]
            i = slot if slot < len(self.tables) else 0
            self.recv = self.recvs[i]
            if self.inline:
                # stream order does everything: the table is complete when the batch's kernels are, and the next batch that
                # writes this table is queued behind the collective -- PROVIDED the detector launched that batch on the stream
                # the collective is queued on (torch's current stream).  On its own stream the next submission could repack
                # the table under the collective: refuse instead of racing.
                cur = torch.cuda.current_stream(self.device)
                used = getattr(self.detector, "last_stream", cur.cuda_stream)
                if used != cur.cuda_stream:
                    raise RuntimeError("PoseGather(inline=True): the detector ran on stream %r, the collective is queued on torch's current stream %r -- "
                                       "pass stream=gather.stream to detect() / submit(), or use inline=False" % (used, cur.cuda_stream))
                e0 = torch.cuda.Event(enable_timing=True); e0.record(cur)
                self.dist.all_gather_into_tensor(self.recv, t)
                e1 = torch.cuda.Event(enable_timing=True); e1.record(cur)
                self._gather_ev.append((e0, e1))
                self.done[i] = e1
                return self.count() if count else -1
            if self.side is None:
                return self.run_table(t) if count else (self.dist.all_gather_into_tensor(self.recv, t), -1)[1]
            # the table is complete (the caller has collected the batch), so the side stream has nothing to wait for
            with torch.cuda.stream(self.side):
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record(self.side)
                self.dist.all_gather_into_tensor(self.recv, t)
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(self.side)
                self._gather_ev.append((e0, ev))
                self.done[i] = ev
                self.last_done = ev
                n = int((self.recv[:, 0] > 0.5).sum().item()) if count else -1
            return n
        return self.run(dets, self.frame_offset)

    def count(self):
        """valid records in the gathered buffer (a host synchronisation on the current stream)"""
        buf = self.recv if self.recv is not None else self.tables[0]
        return int((buf[:, 0] > 0.5).sum().item())

    def before_submit(self, slot):
        """call before the detector is handed a batch whose records go to table `slot`: orders that batch (on the
        detector's stream = torch's current stream) behind the gather that last read the table -- a device-side wait,
        long satisfied by then"""
        if self.done[slot & 1] is not None and getattr(self.device, "type", "cpu") == "cuda":
            torch.cuda.current_stream(self.device).wait_event(self.done[slot & 1])

    @property
    def stream(self):
        """raw HIP stream handle the detector should launch on when its tables feed the collective (torch's current
        stream: RCCL orders itself with it), or None"""
        if self.attached and getattr(self.device, "type", "cpu") == "cuda":
            return torch.cuda.current_stream(self.device).cuda_stream
        return None

    def run_table(self, table):
        """exchange a table that already sits on the device (filled by the detector); returns the number of valid
        records visible to this rank"""
        if self.dist is None:
            return int((table[:, 0] > 0.5).sum().item())
        import time
        t0 = time.perf_counter()
        self.dist.all_gather_into_tensor(self.recv, table)
        n = int((self.recv[:, 0] > 0.5).sum().item())      # also the synchronisation that ends the collective on a GPU rank
        self._gather_s.append(time.perf_counter() - t0)
        return n

    def reset_timing(self):
        self._gather_ev, self._gather_s = [], []

    def gather_ms(self):
        """mean duration of the collectives since reset_timing() in ms (None if there were none): HIP events around the
        all_gather on the side stream for GPU ranks (call after a synchronisation), wall clock around the blocking call
        for CPU ranks and the synchronous GPU form"""
        ms = [a.elapsed_time(b) for a, b in self._gather_ev] + [1e3 * s for s in self._gather_s]
        return (sum(ms) / len(ms)) if ms else None

    def run(self, dets, frame_offset=0):
        """host records -> table -> exchange (CPU ranks; GPU ranks use attach() + run_table())"""
        t = self.tables[0]
        t.copy_(torch.from_numpy(pack(dets, self.nframes, self.tpf, frame_offset)))
        return self.run_table(t)

    def gathered(self, slot=None):
        """The gathered records (world x nslots x REC) of the last exchange, or of the last exchange of result slot `slot`.  GPU ranks: the collective ran on the side stream, so torch's
        current stream is first made to wait for the last gather (a device-side wait) -- a reader on that stream, or a
        .cpu() issued from it, then sees the complete buffer.  One receive buffer per result slot: a batch's records stay until
        the next batch of the SAME slot is exchanged."""
        if self.recv is None:
            return self.tables[0]
        if slot is not None:
            return self.recvs[slot if slot < len(self.recvs) else 0]
        if self.side is not None:
            last = getattr(self, "last_done", None)
            if last is not None:
                torch.cuda.current_stream(self.device).wait_event(last)
        return self.recv
