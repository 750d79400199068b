"""Frame sharding across the GPUs of one node and the single collective of the
path: a variable-length all-gather of box rows.

Frames are independent units (the reference's loop carries no state,
tools/PENet/main.py:238-349); rank r of R takes frames r, r+R, r+2R, ... -- the
interleave of the reference's own DistributedSampler
(pcdet/datasets/__init__.py:48).  Box rows are gathered with the
gather-sizes / pad / gather idiom of pcdet/utils/commu_utils.py:50-100, as one
RCCL all_gather over xGMI (backend "nccl" on ROCm) or gloo on CPU.
"""
import os
from typing import List

import torch
import torch.distributed as dist


def shard_frames(n_frames: int, rank: int, world: int) -> List[int]:
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return list(range(rank, n_frames, world))


def init_from_env(backend=None):
    """torchrun-style env (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def allgather_rows(rows: torch.Tensor, frame_of_view=None) -> torch.Tensor:
    """rows (n_r, C) on every rank -> (sum n_r, C) on every rank, rank-major.

    Two collectives: one all_gather of the int64 row count, one all_gather of
    the rows padded to the maximum count."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return rows
    world = dist.get_world_size()
    cdev = torch.device("cpu") if dist.get_backend() == "gloo" else rows.device
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=cdev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    mx = max(max(counts), 1)
    pad = torch.zeros((mx, rows.shape[1]), dtype=rows.dtype, device=rows.device)
    pad[:rows.shape[0]] = rows
    if dist.get_backend() == "gloo" and pad.is_cuda:      # CPU rehearsal of the multi-GPU path
        cpu = pad.cpu()
        out = [torch.empty_like(cpu) for _ in range(world)]
        dist.all_gather(out, cpu)
        out = [o.to(rows.device) for o in out]
    else:
        out = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(out, pad)
    return torch.cat([o[:c] for o, c in zip(out, counts)], 0)
