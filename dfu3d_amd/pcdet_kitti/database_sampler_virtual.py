"""Row f-4: `la_sampling` of OpenPCDet's ground-truth sampling augmentor for virtual points
(pcdet/datasets/augmentor/database_sampler_virtual.py:307-351), as a batched GPU operation.

The reference bins one sampled object's float32 points by (theta // vert_res, fan // hor_res) in a Python loop over the
points (a dict keyed by strings), keeps the min-theta point of every bin and returns them in first-seen order, or the
object unchanged when fewer than five would remain.  `la_sampling_batch` does the objects of a frame in one launch
(`dfu3d_la_sampling`: one workgroup per object); `la_sampling` keeps the reference's one-object signature."""
import ctypes

import numpy as np
import torch

from .. import _lib
from ..stages import _chk, _stream


def la_sampling_batch(objects, vert_res=0.006, hor_res=0.003, device="cuda:0"):
    """objects: list of float32 arrays / tensors (n_i, C >= 3), x y z first -> list of sampled arrays (NumPy in -> NumPy
    out, tensor in -> tensor on `device`).  Same rows, same order as the reference's la_sampling per object."""
    if not objects:
        return []
    as_numpy = not torch.is_tensor(objects[0])
    ts = [torch.as_tensor(np.ascontiguousarray(o, np.float32)) if not torch.is_tensor(o) else o.to(torch.float32) for o in objects]
    C = int(ts[0].shape[1])
    if C < 3 or any(t.dim() != 2 or int(t.shape[1]) != C for t in ts):
        raise _lib.Dfu3dError("la_sampling: every object must be (n, C) with the same C >= 3")
    counts = [int(t.shape[0]) for t in ts]
    n = sum(counts)
    if n == 0:
        return [o for o in objects]
    pts = torch.cat([t.to(device) for t in ts], 0).contiguous()
    off = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int64, device=device)
    out = torch.empty_like(pts)
    cnt = torch.zeros(len(ts), dtype=torch.int32, device=device)
    scratch = torch.empty(3 * n, dtype=torch.int64, device=device)            # 24 bytes per point
    rc = _lib.lib().dfu3d_la_sampling(_chk(pts, "points", torch.float32, numel=n * C), C,
                                      _chk(off, "obj_off", torch.int64, numel=len(ts) + 1), len(ts),
                                      ctypes.c_float(np.float32(vert_res)), ctypes.c_float(np.float32(hor_res)),
                                      _chk(out, "out", torch.float32, numel=n * C),
                                      _chk(cnt, "out_cnt", torch.int32, numel=len(ts)),
                                      _chk(scratch, "scratch", torch.int64, numel=3 * n), n, _stream())
    _lib.check(rc, "dfu3d_la_sampling")
    k = cnt.cpu().tolist()
    o0 = np.concatenate([[0], np.cumsum(counts)])
    res = [out[int(o0[b]):int(o0[b]) + k[b]] for b in range(len(ts))]
    return [r.cpu().numpy() for r in res] if as_numpy else res


def la_sampling(points, vert_res=0.006, hor_res=0.003, device="cuda:0"):
    """database_sampler_virtual.py:319: one object."""
    return la_sampling_batch([points], vert_res, hor_res, device)[0]
