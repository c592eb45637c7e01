"""Multi-rank path on CPU: frame sharding + the one all_gather of pose records (gloo, world 2), the step loop of
bench.py driven by a stand-in detector on two CPU ranks, and bench.py's refusal to run a smaller world than asked for."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from robot_camera_calibration_amd import api, dist as rdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _fake_dets(rank, nframes, tpf=1):
    """deterministic stand-ins for rcc_detection records: every third frame of a rank has no target; with tpf > 1 frame
    f carries 1 + (f % tpf) tags"""
    rows = []
    for f in range(nframes):
        if (f + rank) % 3 == 0:
            continue
        for q in range(1 if tpf == 1 else 1 + (f % tpf)):
            rows.append((f, q))
    d = np.zeros(len(rows), api.DET_DT).view(np.recarray)
    fr = np.array([r[0] for r in rows]); qq = np.array([r[1] for r in rows])
    d.frame = fr
    d.id = qq
    d.ncorners = 48 if tpf == 1 else 4
    d.rvec = np.stack([rank + 0.1 * fr, 0.2 + qq, 0.3 + 0 * fr], 1)
    d.tvec = np.stack([0.0 * fr, rank + 0.0 * fr, fr + 0.01 * qq], 1)
    d.rms = 0.05
    d.corners = np.arange(8, dtype=float).reshape(4, 2)[None] + (100.0 * fr + qq)[:, None, None]
    return d


def _check_block(blk, d, nframes, tpf, frame_offset):
    """a rank's (nframes * tpf, REC) block against the records it was packed from -- every field, all four corners"""
    assert blk.shape == (nframes * tpf, rdist.REC) and rdist.REC == 19
    first = np.searchsorted(d.frame, d.frame, side="left")
    slot = d.frame * tpf + (np.arange(len(d)) - first)
    valid = blk[:, 0] > 0.5
    assert sorted(np.flatnonzero(valid)) == sorted(slot)
    assert np.array_equal(blk[slot, 1], d.frame + frame_offset)                      # global frame index
    assert np.array_equal(blk[slot, 2], d.id) and np.array_equal(blk[slot, 3], d.ncorners)
    assert np.array_equal(blk[slot, 4:7], d.rvec) and np.array_equal(blk[slot, 7:10], d.tvec)
    assert np.array_equal(blk[slot, 10], d.rms)
    assert np.array_equal(blk[slot, 11:19], d.corners.reshape(len(d), 8))             # bl, br, tr AND tl
    assert np.array_equal(blk[slot, 17:19], d.corners[:, 3, :])                       # the 4th corner (corner_detections.cpp:51-56 reads four)
    assert (blk[~valid] == 0).all()


def _worker(rank, world, port, nframes, tpf, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = rdist.PoseGather(nframes, torch.device("cpu"), world, dist, rank, targets_per_frame=tpf)
    n = g.run(_fake_dets(rank, nframes, tpf), frame_offset=rank * nframes)
    out = g.gathered().numpy().copy()
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, n, out))


@pytest.mark.parametrize("tpf", [1, 3])
def test_pose_gather_world2(tpf):
    world, nframes = 2, 12
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, nframes, tpf, q)) for r in range(world)]
    for p in ps: p.start()
    res = sorted([q.get(timeout=120) for _ in ps], key=lambda t: t[0])
    for p in ps: p.join(30)
    exp_valid = sum(len(_fake_dets(r, nframes, tpf)) for r in range(world))
    nslots = nframes * tpf
    for rank, n, out in res:
        assert n == exp_valid and out.shape == (world * nslots, rdist.REC)
        for r in range(world):
            _check_block(out[r * nslots:(r + 1) * nslots], _fake_dets(r, nframes, tpf), nframes, tpf, r * nframes)
    assert np.array_equal(res[0][2], res[1][2])                                       # every rank sees the same table


def _ragged_worker(rank, world, port, nframes, short_rank, short_n, q):
    """three exchanges: a full batch, then a batch that is SHORT on one rank (the ragged end of a stream: its table's remaining
    slots must be zero, not the previous batch's records), then a full one again"""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = rdist.PoseGather(nframes, torch.device("cpu"), world, dist, rank)
    outs = []
    for step in range(3):
        n_here = short_n if (step == 1 and rank == short_rank) else nframes
        d = _fake_dets(rank + 7 * step, n_here)
        cnt = g.run(d, frame_offset=rank * nframes)
        outs.append((cnt, g.gathered().numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, outs))


def test_pose_gather_world4_ragged_last_batch():
    """the first real 8-GPU run will be the first time more than two ranks meet: four gloo ranks, one of
    which ends its stream with a short batch -- every rank sees every rank's records in its block, the short rank's unused slots
    are zero in that step (no stale records of the step before), and the next full step is whole again"""
    world, nframes, short_rank, short_n = 4, 10, 2, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_ragged_worker, args=(r, world, port, nframes, short_rank, short_n, q)) for r in range(world)]
    for p in ps: p.start()
    res = sorted([q.get(timeout=180) for _ in ps], key=lambda t: t[0])
    for p in ps: p.join(30)
    for step in range(3):
        exp = 0
        for r in range(world):
            n_here = short_n if (step == 1 and r == short_rank) else nframes
            exp += len(_fake_dets(r + 7 * step, n_here))
        for rank, outs in res:
            cnt, tab = outs[step]
            assert cnt == exp and tab.shape == (world * nframes, rdist.REC)
            for r in range(world):
                n_here = short_n if (step == 1 and r == short_rank) else nframes
                blk = tab[r * nframes:(r + 1) * nframes]
                _check_block(blk, _fake_dets(r + 7 * step, n_here), nframes, 1, r * nframes)
                assert (blk[n_here:] == 0).all()
        assert all(np.array_equal(res[0][1][step][1], o[1][step][1]) for o in res[1:])


def test_pack_refuses_to_truncate():
    d = _fake_dets(0, 12, tpf=3)
    with pytest.raises(ValueError):
        rdist.pack(d, 12, targets_per_frame=2)        # a frame with 3 tags does not fit 2 slots: an error, not a cut
    with pytest.raises(ValueError):
        rdist.pack(d, 6, targets_per_frame=3)         # frame index beyond the batch
    with pytest.raises(ValueError):
        rdist.pack(d[::-1], 12, targets_per_frame=3)  # not ordered by frame
    assert rdist.pack(d[:0], 12, 3).shape == (36, rdist.REC)


def test_shard_range_partitions_frames():
    for n in (0, 1, 7, 1024, 1031):
        for w in (1, 2, 3, 8):
            r = [rdist.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(hi - lo for lo, hi in r) - min(hi - lo for lo, hi in r) <= 1


def test_world1_needs_no_collective():
    g = rdist.PoseGather(8, torch.device("cpu"), 1, None)
    assert g.exchange(_fake_dets(0, 8)) == len(_fake_dets(0, 8))


class _StandInDetector:
    """detect / submit / collect with the calling convention of api.Detector, producing _fake_dets: lets bench.py's
    own step loop run on CPU ranks (there is no GPU in the CPU test box)"""

    def __init__(self, rank, nframes):
        self.rank, self.nframes, self._pending, self._nsub, self.calls = rank, nframes, [], 0, []

    def detect(self, frames, n, want_corners=False, stream=None):
        self.calls.append("detect")
        return _fake_dets(self.rank, n), None

    def submit(self, frames, n, stream=None):
        assert len(self._pending) < 2, "more than two submissions outstanding"
        self.calls.append("submit")
        self._pending.append((n, self._nsub & 1)); self._nsub += 1

    def collect(self):
        self.calls.append("collect")
        n, self.last_slot = self._pending.pop(0)
        return _fake_dets(self.rank, n), None


def _bench_worker(rank, world, port, nframes, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = rdist.PoseGather(nframes, torch.device("cpu"), world, dist, rank)
    g.frame_offset = rank * nframes
    det = _StandInDetector(rank, nframes)
    import time
    g.reset_timing()
    t0 = time.perf_counter()
    found_stream = bench.run_steps(det, None, nframes, g, 3, False)
    dt = time.perf_counter() - t0
    tab = g.gathered().numpy().copy()
    # the fields that make an N > 1 line attributable: every rank contributes, every rank gets all of them
    rep = bench.rank_report(dist, torch.device("cpu"), world, dt, 3, bench.run_steps.local_found, g, {"numa_node": rank, "pci": "x", "cpus": 1} if rank == 1 else None)
    found_sync = bench.run_steps(det, None, nframes, g, 2, True)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, found_stream, found_sync, tab, det.calls, rep))


def test_bench_step_loop_world2():
    """bench.run_steps -- the body of bench.py's timed region -- on two gloo ranks: every step ends in one all_gather,
    the streaming form keeps one batch ahead, and both ranks see both ranks' records"""
    world, nframes = 2, 9
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_bench_worker, args=(r, world, port, nframes, q)) for r in range(world)]
    for p in ps: p.start()
    res = sorted([q.get(timeout=120) for _ in ps], key=lambda t: t[0])
    for p in ps: p.join(30)
    exp = sum(len(_fake_dets(r, nframes)) for r in range(world))
    for rank, fs, fy, tab, calls, rep in res:
        assert fs == exp and fy == exp
        assert rep["collective_world"] == world and rep["collective_backend"] == "gloo"
        assert rep["found_per_rank"] == [len(_fake_dets(r, nframes)) for r in range(world)] and sum(rep["found_per_rank"]) == exp
        assert len(rep["per_rank_ms_per_step"]) == world and all(v > 0 for v in rep["per_rank_ms_per_step"])
        assert len(rep["gather_ms_per_step"]) == world and all(v is not None and v > 0 for v in rep["gather_ms_per_step"])   # 3 collectives timed per rank
        assert rep["ranks_pinned_to_numa"] == [False, True] and rep["numa_node_per_rank"] == [-1, 1]
        assert calls[:7] == ["submit", "submit", "collect", "submit", "collect", "collect", "detect"]
        for r in range(world):
            _check_block(tab[r * nframes:(r + 1) * nframes], _fake_dets(r, nframes), nframes, 1, r * nframes)
    assert np.array_equal(res[0][3], res[1][3])


def test_bench_refuses_a_smaller_world():
    """`bench.py --gpus 2` where fewer than two devices are visible must fail loudly, not print n_gpus: 1"""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    if torch.cuda.device_count() >= 2:
        pytest.skip("two devices visible: the launcher would really start two ranks")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "n_gpus" not in r.stdout
    assert "--gpus 2" in r.stderr
    # the launcher parent counts devices from sysfs: it must get to its verdict without importing torch (no HIP call)
    code = ("import sys; sys.path.insert(0, %r); sys.argv = ['bench.py', '--gpus', '2']; import bench; rc = bench.main(); "
            "assert 'torch' not in sys.modules, 'the launcher parent imported torch'; sys.exit(rc)" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2 and "AssertionError" not in r.stderr, r.stderr
    # launched as one rank of a world that does not match --gpus: an error as well
    env["WORLD_SIZE"], env["RANK"], env["LOCAL_RANK"] = "1", "0", "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "n_gpus" not in r.stdout
