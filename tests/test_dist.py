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


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from dfu3d_amd import dist as D
    import torch.distributed as dist
    D.init_from_env("gloo")
    n = 3 + 4 * rank if rank != 1 else 0            # ragged, one empty rank
    rows = torch.arange(n * 18, dtype=torch.float64).reshape(n, 18) + 1000 * rank
    out = D.allgather_rows(rows)
    q.put((rank, out.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_allgather_rows_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
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
