"""Multi-GPU: frame sharding and the one collective of the path.

Frames are independent units (the tag-map accumulation of real_preprocessing/src/camera_pose.cpp:176-203
is downstream of this path), so N ranks simply own disjoint frame ranges and no collective touches
pixels.  The only exchange is the per-batch all-gather of fixed-size pose records so that one
process (the ROS node's publisher, or the map builder) sees every frame's result -- the role the
"tag_detections" topic plays between processes in the reference (corner_detections.cpp:78).
Records are 17 doubles per frame slot (~139 KB per rank for 1024 frames): latency-bound on xGMI,
so it is ONE all_gather per batch, never per frame.
"""
import numpy as np
import torch

REC = 17  # valid, frame, id, ncorners, rvec[3], tvec[3], rms, corners bl/br/tr/tl x,y (first 6 of 8 kept) -> see pack()


def shard_range(nframes, rank, world):
    """Contiguous block partition of frame indices: rank r owns [lo, hi)."""
    base, rem = divmod(nframes, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def pack(dets, nslots, frame_offset=0):
    """rcc_detection records -> (nslots, REC) float64, zero-padded; slot = frame index in the batch."""
    a = np.zeros((nslots, REC), np.float64)
    if len(dets) == 0:
        return a
    d = np.asarray(dets)
    if len(d) > nslots or len(np.unique(d["frame"])) != len(d):
        # several targets per frame (fiducials): slot = running index, capped at nslots
        d = d[:nslots]
        fr = np.arange(len(d))
        a[fr, 0] = 1.0
        a[fr, 1] = d["frame"] + frame_offset
        a[fr, 2] = d["id"]; a[fr, 3] = d["ncorners"]; a[fr, 4:7] = d["rvec"]; a[fr, 7:10] = d["tvec"]; a[fr, 10] = d["rms"]
        a[fr, 11:17] = d["corners"].reshape(len(d), 8)[:, :6]
        return a
    fr = d["frame"]
    a[fr, 0] = 1.0
    a[fr, 1] = fr + frame_offset
    a[fr, 2] = d["id"]
    a[fr, 3] = d["ncorners"]
    a[fr, 4:7] = d["rvec"]
    a[fr, 7:10] = d["tvec"]
    a[fr, 10] = d["rms"]
    a[fr, 11:17] = d["corners"].reshape(len(d), 8)[:, :6]
    return a


class PoseGather:
    """One all_gather of pose records per batch.  world == 1: no collective at all."""

    def __init__(self, nslots, device, world, dist_module=None, rank=0):
        self.nslots, self.device, self.world, self.dist, self.rank = nslots, device, world, dist_module, rank
        self.send = torch.zeros((nslots, REC), dtype=torch.float64, device=device)
        self.recv = torch.zeros((world * nslots, REC), dtype=torch.float64, device=device) if world > 1 else None
        self.pinned = torch.zeros((nslots, REC), dtype=torch.float64)
        if torch.cuda.is_available() and getattr(device, "type", "cpu") == "cuda":
            self.pinned = self.pinned.pin_memory()

    def run(self, dets, frame_offset=0):
        """returns the number of valid records visible to this rank after the exchange"""
        if self.world == 1 or self.dist is None:
            return len(dets)
        self.pinned.numpy()[:] = pack(dets, self.nslots, frame_offset)
        self.send.copy_(self.pinned, non_blocking=True)
        self.dist.all_gather_into_tensor(self.recv, self.send)
        return int((self.recv[:, 0] > 0.5).sum().item())

    def gathered(self):
        return self.recv if self.recv is not None else self.send
