"""Reader PROCESSES for the batched directory labeler ("pinned-memory async H2D +
parallel decode").  Python threads top out near 1 ms of GIL-bound work per frame (calibration parse, zip
directory, array headers) and contend on the address space when they allocate 17 MB per frame; forked
workers have neither problem.  Each worker reads a frame with pipeline.read_frame and copies the payloads
into a POSIX shared-memory staging area at a fixed per-frame stride; the parent page-locks that area once
(hipHostRegister) and uploads from it.

The pool must be created BEFORE the process initialises the GPU (a forked child of a GPU process would
inherit its device handles); ReaderPool() refuses otherwise and the labeler falls back to threads.
"""
import multiprocessing as mp
import os
from multiprocessing import shared_memory

import numpy as np


def _attach(cache, name):
    shm = cache.get(name)
    if shm is None:
        shm = shared_memory.SharedMemory(name=name)
        try:                                    # the parent owns the segment: keep this process's resource
            from multiprocessing import resource_tracker      # tracker from unlinking it when the worker exits
            resource_tracker.unregister(shm._name, "shared_memory")
        except Exception:
            pass
        cache[name] = shm
    return shm


def _worker(tasks, done):
    from .pipeline import read_frame          # imported in the child: no torch.cuda use below
    cache = {}
    while True:
        t = tasks.get()
        if t is None:
            break
        (tag, i, root, stem, depth_dir, seg_dir, want_image, shm_name, lay) = t
        try:
            f = read_frame(root, stem, depth_dir, want_image, seg_dir=seg_dir)
            buf = _attach(cache, shm_name).buf
            H, W = f.depth.shape
            n, m = f.points.shape[0], min(f.masks.shape[0], lay["max_masks"])
            if (H, W) != (lay["H"], lay["W"]):
                raise ValueError("%s: depth %s differs from the batch's %s" % (stem, (H, W), (lay["H"], lay["W"])))
            if n > lay["cap_n"]:
                raise ValueError("%s: %d LiDAR points exceed cap_n=%d" % (stem, n, lay["cap_n"]))
            view = lambda off, shape, dt: np.ndarray(shape, dt, buffer=buf, offset=off)
            np.copyto(view(lay["pts"] + i * lay["cap_n"] * 16, (n, 4), np.float32), f.points)
            np.copyto(view(lay["depth"] + i * H * W * 4, (H, W), np.float32), f.depth)
            if m:
                np.copyto(view(lay["masks"] + i * lay["max_masks"] * H * W, (m, H, W), np.uint8), f.masks[:m])
            if want_image:
                np.copyto(view(lay["image"] + i * H * W * 3, (H, W, 3), np.uint8), f.image)
            done.put((tag, i, None, dict(stem=stem, n=n, m=m, m_total=int(f.masks.shape[0]), classes=f.classes[:m], scores=f.scores[:m],
                                         boxes=f.boxes[:m], thing_classes=f.thing_classes,
                                         calib=f.calib.record())))
        except Exception as e:                       # reported to the parent, the worker lives on
            done.put((tag, i, "%s: %s" % (type(e).__name__, e), None))


class ReaderPool:
    def __init__(self, workers=8):
        try:
            import torch
            if torch.cuda.is_initialized():
                raise RuntimeError("ReaderPool must be created before the GPU is initialised")
        except ImportError:
            pass
        ctx = mp.get_context("fork")
        self.tasks, self.done = ctx.Queue(), ctx.Queue()
        self.procs = [ctx.Process(target=_worker, args=(self.tasks, self.done), daemon=True) for _ in range(workers)]
        for p in self.procs:
            p.start()
        self._tag = 0

    @staticmethod
    def layout(B, H, W, cap_n, max_masks, want_image):
        """Byte offsets of the staging planes for B frames; total size under key 'bytes'."""
        lay = dict(B=B, H=H, W=W, cap_n=cap_n, max_masks=max_masks)
        off = 0
        for key, nbytes in (("pts", B * cap_n * 16), ("depth", B * H * W * 4), ("masks", B * max_masks * H * W),
                            ("image", B * H * W * 3 if want_image else 0)):
            lay[key] = off
            off += (nbytes + 4095) // 4096 * 4096
        lay["bytes"] = max(off, 4096)
        return lay

    def read_batch(self, root, stems, depth_dir, seg_dir, want_image, shm_name, lay):
        """Blocks until every frame of the batch is in the staging area -> list of per-frame metadata dicts."""
        self._tag += 1
        for i, s in enumerate(stems):
            self.tasks.put((self._tag, i, root, s, depth_dir, seg_dir, want_image, shm_name, lay))
        out = [None] * len(stems)
        for _ in stems:
            tag, i, err, meta = self.done.get()
            if tag != self._tag:
                raise RuntimeError("reader pool: stale reply")
            if err:
                raise RuntimeError("reader: " + err)
            out[i] = meta
        return out

    def close(self):
        for _ in self.procs:
            self.tasks.put(None)
        for p in self.procs:
            p.join(timeout=5)
            if p.is_alive():
                p.terminate()
        self.procs = []
