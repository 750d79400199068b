"""Frame sharding across the GPUs of one node and the single collective of the
path: a variable-length all-gather of box rows.

Frames are independent units (the reference's loop carries no state,
tools/PENet/main.py:238-349); rank r of R takes frames r, r+R, r+2R, ... -- the
interleave of the reference's own DistributedSampler
(pcdet/datasets/__init__.py:48).  Box rows are gathered as ONE RCCL all_gather per step over xGMI
(backend "nccl" on ROCm; gloo on CPU) of fixed-capacity buffers that carry their own row count; a one-off gather
of a whole run's rows uses the gather-sizes / pad / gather idiom of pcdet/utils/commu_utils.py:50-100.
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


# what allgather_rows has cost so far in this process (tests/test_dist.py asserts the per-step figures)
STATS = {"collectives": 0, "host_syncs": 0}


def allgather_rows(rows: torch.Tensor, cap_rows: int = None) -> torch.Tensor:
    """rows (n_r, C) on every rank -> (sum n_r, C) on every rank, rank-major.

    cap_rows given (the per-step path: bench.py, a labelling loop): ONE collective and ONE host synchronisation per
    call.  Every rank sends a fixed-capacity (cap_rows + 1, C) buffer whose first row carries its row count -- no
    count exchange in front, no `.item()` per rank; the counts of all ranks are read back together after the gather.
    cap_rows must be the same on every rank and an upper bound of every rank's n_r (the engine's row capacity: more
    rows cannot exist); a rank with more raises on EVERY rank, after the collective, so that nobody hangs.
    cap_rows None (a one-off gather of a whole run's rows): the gather-sizes / pad / gather idiom of
    pcdet/utils/commu_utils.py:50-100 -- two collectives, world + 1 host synchronisations."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return rows
    world = dist.get_world_size()
    gloo = dist.get_backend() == "gloo"
    C = rows.shape[1]
    if cap_rows is not None:
        cap = int(cap_rows)
        cdev = torch.device("cpu") if gloo else rows.device          # (gloo: CPU rehearsal of the multi-GPU path)
        buf = torch.zeros((cap + 1, C), dtype=rows.dtype, device=rows.device)
        n = min(rows.shape[0], cap)
        buf[0, 0] = rows.shape[0]                                     # the TRUE count: an overflow is seen by everyone
        buf[1:1 + n] = rows[:n]
        send = buf.to(cdev)
        out = torch.empty((world * (cap + 1), C), dtype=rows.dtype, device=cdev)
        dist.all_gather_into_tensor(out, send)
        STATS["collectives"] += 1
        out = out.view(world, cap + 1, C)
        counts = out[:, 0, 0].to("cpu").tolist()                      # the one host synchronisation
        STATS["host_syncs"] += 1
        counts = [int(c) for c in counts]
        if max(counts) > cap:
            raise RuntimeError("allgather_rows: rank %d holds %d rows, capacity %d" % (int(np.argmax(counts)), max(counts), cap))
        return torch.cat([out[r, 1:1 + c] for r, c in enumerate(counts)], 0).to(rows.device)
    cdev = torch.device("cpu") if gloo else rows.device
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=cdev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    STATS["collectives"] += 1
    counts = [int(c.item()) for c in counts]
    STATS["host_syncs"] += world
    mx = max(max(counts), 1)
    pad = torch.zeros((mx, C), dtype=rows.dtype, device=rows.device)
    pad[:rows.shape[0]] = rows
    if gloo and pad.is_cuda:      # CPU rehearsal of the multi-GPU path
        cpu = pad.cpu()
        out = [torch.empty_like(cpu) for _ in range(world)]
        dist.all_gather(out, cpu)
        out = [o.to(rows.device) for o in out]
    else:
        out = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(out, pad)
    STATS["collectives"] += 1
    STATS["host_syncs"] += 1
    return torch.cat([o[:c] for o, c in zip(out, counts)], 0)
