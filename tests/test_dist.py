"""CPU-only: frame sharding and the row all-gather over gloo, world_size 2."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def test_shard_frames_partitions_all_frames():
    from dfu3d_amd.dist import shard_frames
    for world in (1, 2, 4, 8):
        got = sorted(sum((shard_frames(28130, r, world) for r in range(world)), []))
        assert got == list(range(28130))
        sizes = [len(shard_frames(28130, r, world)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1
    assert shard_frames(10, 1, 4) == [1, 5, 9]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, cap):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from dfu3d_amd import dist as D
    import torch.distributed as dist
    D.init_from_env("gloo")
    n = 3 + 4 * rank if rank != 1 else 0            # ragged, one empty rank
    rows = torch.arange(n * 18, dtype=torch.float64).reshape(n, 18) + 1000 * rank
    before = dict(D.STATS)
    out = D.allgather_rows(rows, cap_rows=cap)
    cost = {k: D.STATS[k] - before[k] for k in before}
    err = None
    if cap is not None:                             # a rank over the capacity: everybody raises, after the collective
        try:
            D.allgather_rows(torch.zeros((cap + 1 if rank == world - 1 else 1, 18), dtype=torch.float64), cap_rows=cap)
        except RuntimeError as e:
            err = str(e)
    q.put((rank, out.numpy(), cost, err))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,cap", [(2, 16), (3, 16), (2, None), (3, None)])
def test_allgather_rows_gloo(world, cap):
    """cap given (the per-step path): ONE collective and ONE host synchronisation per call whatever the number of
    ranks -- at N = 8 the count exchange + one `.item()` per rank of the two-collective form were ten host
    synchronisations per 7.7 ms step; cap None: that form, still used for the one-off gather at the end of a run."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, cap)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    res = {r: o for r, o, _, _ in got}
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    exp = []
    for r in range(world):
        n = 3 + 4 * r if r != 1 else 0
        exp.append(np.arange(n * 18, dtype=np.float64).reshape(n, 18) + 1000 * r)
    exp = np.concatenate(exp)
    for r in range(world):
        assert np.array_equal(res[r], exp)
    for r, _, cost, err in got:
        if cap is not None:
            assert cost == {"collectives": 1, "host_syncs": 1}, (r, cost)
            assert err is not None and "capacity" in err, (r, err)
        else:
            assert cost == {"collectives": 2, "host_syncs": world + 1}, (r, cost)


# ------------------------------------------------------------------ rows carry their frame
def _fake_engine_rows(rng, n_views, per_view):
    """(n,24) rows as the engine emits them: col 0 view, 1 inst, 2 cluster, 3 class, 4.. numbers."""
    rows = []
    for v in range(n_views):
        for k in range(per_view[v]):
            r = rng.normal(0, 10, 24)
            r[0], r[1], r[2], r[3] = v, k // 2, k % 2, rng.integers(0, 10)
            rows.append(r)
    return torch.tensor(np.array(rows).reshape(-1, 24), dtype=torch.float64)


def _worker_frames(rank, world, port, q, n_frames, cams):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from dfu3d_amd import dist as D
    import torch.distributed as dist
    D.init_from_env("gloo")
    mine = D.shard_frames(n_frames, rank, world)              # global frame ids of this rank
    view_frame = np.repeat(np.arange(len(mine)), cams)        # batch-local frame of every view
    view_cam = np.tile(np.arange(cams), len(mine))
    rng = np.random.default_rng(100 + rank)
    per_view = rng.integers(0, 4, len(view_frame))
    local = _fake_engine_rows(rng, len(view_frame), per_view)
    out = D.allgather_rows(D.rows_for_gather(local, view_frame, mine, view_cam))
    q.put((rank, out.numpy(), local.numpy(), np.array(mine)))
    dist.barrier()
    dist.destroy_process_group()


def test_gathered_rows_are_attributable_to_frames(tmp_path):
    """Every gathered row carries the GLOBAL frame and the camera it came from: rows of rank r belong to
    frames r, r+R, ... (the reference's DistributedSampler interleave), (frame, cam, inst, cluster) is a key
    of the gathered list, and the numbers survive the trip."""
    world, n_frames, cams = 2, 7, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_frames, args=(r, world, port, q, n_frames, cams)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r: (g, loc, mine) for r, g, loc, mine in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    g0 = res[0][0]
    assert all(np.array_equal(res[r][0], g0) for r in range(world))          # the same list on every rank
    assert g0.shape[1] == 18
    # rank-major: rows of rank r come r-th and all of them belong to r's shard
    start = 0
    for r in range(world):
        loc, mine = res[r][1], res[r][2]
        part = g0[start:start + loc.shape[0]]
        start += loc.shape[0]
        assert set(part[:, 0].astype(int)) <= set(mine.tolist())
        assert np.all(part[:, 0].astype(int) % world == r)
        v = loc[:, 0].astype(int)
        assert np.array_equal(part[:, 0].astype(int), mine[v // cams]) and np.array_equal(part[:, 1].astype(int), v % cams)
        assert np.array_equal(part[:, 2], loc[:, 3]) and np.array_equal(part[:, 3:5], loc[:, 1:3])
        assert np.array_equal(part[:, 5:18], loc[:, 4:17])
    assert start == g0.shape[0]
    keys = {tuple(k) for k in g0[:, [0, 1, 3, 4]].astype(int).tolist()}
    assert len(keys) == g0.shape[0]                                         # no (frame, cam, inst, cluster) collision
    from dfu3d_amd.dist import write_manifest
    man = write_manifest(str(tmp_path / "m.json"), g0, n_frames, world)
    assert man["boxes"] == g0.shape[0] and sum(man["boxes_per_rank"]) == g0.shape[0]
    assert man["boxes_per_rank"] == [res[r][1].shape[0] for r in range(world)]
