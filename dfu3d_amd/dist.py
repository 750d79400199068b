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

import numpy as np
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


GATHER_COLUMNS = ("frame_idx", "cam", "class", "inst", "cluster", "alpha", "x1", "y1", "x2", "y2",
                  "h", "w", "l", "x", "y", "z", "ry", "score")


def rows_for_gather(rows: torch.Tensor, view_frame, frame_ids, view_cam) -> torch.Tensor:
    """Engine rows (n,24; column 0 = view index inside the rank's batch, include/dfu3d.h) -> the rows that
    travel between ranks: (n,18) float64 in GATHER_COLUMNS order.  view_frame (V,) maps a view
    to its frame inside the batch, frame_ids (F,) that frame to its GLOBAL index, view_cam (V,) to its camera;
    afterwards a row can be attributed to its frame whatever rank produced it.  (The box numbers stay
    float64 -- the label files print them at full precision -- instead of 8e's i32/f32 mix.)"""
    d = rows.device
    t = lambda a: a.to(d) if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a), device=d)
    vf, fid, cam = t(view_frame).long(), t(frame_ids).to(torch.float64), t(view_cam).to(torch.float64)
    out = torch.empty((rows.shape[0], len(GATHER_COLUMNS)), dtype=torch.float64, device=d)
    if rows.shape[0]:
        v = rows[:, 0].long().clamp_(0, vf.numel() - 1)     # (no host sync here: this sits on the per-step path)
        out[:, 0] = fid[vf[v]]
        out[:, 1] = cam[v]
        out[:, 2] = rows[:, 3]
        out[:, 3] = rows[:, 1]
        out[:, 4] = rows[:, 2]
        out[:, 5:18] = rows[:, 4:17]
    return out


def write_manifest(path, gathered: torch.Tensor, n_frames_total: int, world: int, extra=None):
    """Rank 0's summary of a sharded run: frames, boxes, boxes per rank (frame_idx % world == rank, the
    interleave of shard_frames) and per class."""
    import json
    g = gathered.cpu().numpy() if isinstance(gathered, torch.Tensor) else np.asarray(gathered)
    fr = g[:, 0].astype(np.int64) if g.shape[0] else np.zeros((0,), np.int64)
    man = {"frames": int(n_frames_total), "world_size": int(world), "boxes": int(g.shape[0]),
           "frames_with_boxes": int(np.unique(fr).size),
           "boxes_per_rank": [int((fr % world == r).sum()) for r in range(world)],
           "boxes_per_class": {str(int(c)): int(n) for c, n in zip(*np.unique(g[:, 2].astype(np.int64), return_counts=True))}
           if g.shape[0] else {},
           "columns": list(GATHER_COLUMNS)}
    if extra:
        man.update(extra)
    with open(path, "w") as f:
        json.dump(man, f, indent=1)
    return man


def allgather_rows(rows: torch.Tensor) -> torch.Tensor:
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
