"""Multi-rank path on CPU: frame sharding + the one all_gather of pose records (gloo, world 2)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from robot_camera_calibration_amd import api, dist as rdist


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _fake_dets(rank, nslots):
    """deterministic stand-ins for rcc_detection records: every third slot of a rank has no board"""
    idx = [f for f in range(nslots) if (f + rank) % 3 != 0]
    d = np.zeros(len(idx), api.DET_DT).view(np.recarray)
    d.frame = idx
    d.id = 0
    d.ncorners = 48
    d.rvec = np.array([[rank + 0.1 * f, 0.2, 0.3] for f in idx])
    d.tvec = np.array([[0.0, rank, f] for f in idx], float)
    d.rms = 0.05
    d.corners = np.arange(8, dtype=float).reshape(4, 2)[None] + np.array(idx)[:, None, None]
    return d


def _worker(rank, world, port, nslots, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = rdist.PoseGather(nslots, torch.device("cpu"), world, dist, rank)
    n = g.run(_fake_dets(rank, nslots), frame_offset=rank * nslots)
    out = g.gathered().numpy().copy()
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, n, out))


def test_pose_gather_world2():
    world, nslots = 2, 12
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, nslots, q)) for r in range(world)]
    for p in ps: p.start()
    res = sorted([q.get(timeout=120) for _ in ps])
    for p in ps: p.join(30)
    exp_valid = sum(len(_fake_dets(r, nslots)) for r in range(world))
    for rank, n, out in res:
        assert n == exp_valid and out.shape == (world * nslots, rdist.REC)
        for r in range(world):
            blk = out[r * nslots:(r + 1) * nslots]
            d = _fake_dets(r, nslots)
            valid = blk[:, 0] > 0.5
            assert sorted(np.flatnonzero(valid)) == sorted(d.frame)
            assert np.array_equal(blk[d.frame, 1], d.frame + r * nslots)          # global frame index
            assert np.array_equal(blk[d.frame, 4:7], d.rvec) and np.array_equal(blk[d.frame, 7:10], d.tvec)
            assert (blk[~valid] == 0).all()
    assert np.array_equal(res[0][2], res[1][2])                                       # every rank sees the same table


def test_shard_range_partitions_frames():
    for n in (0, 1, 7, 1024, 1031):
        for w in (1, 2, 3, 8):
            r = [rdist.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(hi - lo for lo, hi in r) - min(hi - lo for lo, hi in r) <= 1


def test_world1_needs_no_collective():
    g = rdist.PoseGather(8, torch.device("cpu"), 1, None)
    assert g.run(_fake_dets(0, 8)) == len(_fake_dets(0, 8))
