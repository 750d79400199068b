"""Batched labelling of a KITTI-format directory (the L0
readers either side of the hot path).

The reference walks the directory one frame at a time
(tools/PENet/main.py:238-349, vis_utils.py:136-166).  Here frames are read by a
pool of host threads (velodyne/*.bin, calib/*.txt, image_2/*.png, seg_2/*.npz,
depth_2/*.npy; tools/nuscenes2kitti.py:246-285 for the formats), packed into
PINNED staging tensors, copied to the GPU on a side stream while the previous
batch is still being processed, run through PseudoBoxEngine in one multi-view
launch chain, and the label files (and, optionally, the float16 virtual-point
files of vis_utils.py:164-166) are written by background threads.
"""
import os
import time
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch

from .calibration import Calibration
from .engine import PseudoBoxEngine, ViewBatch
from .labels import write_label_file
from .params import NUSC_CLASSES, Params


@dataclass
class Frame:
    stem: str
    points: np.ndarray          # (n,4) f32
    calib: Calibration
    depth: np.ndarray           # (H,W) f32
    masks: np.ndarray           # (M,H,W) u8
    classes: np.ndarray         # (M,) int
    scores: np.ndarray          # (M,) f32
    boxes: np.ndarray           # (M,4) f32
    thing_classes: List[str]
    image: Optional[np.ndarray] = None   # (H,W,3) u8, only for the virtual-point file
    n_total: int = 0                     # instances of the frame (a batch view holds at most 32 of them)


def _npz_arrays(path):
    """np.load(path) for a .npz without pickles, but each member is inflated by ONE zlib call
    (zipfile's chunked reader holds the GIL most of the time, which serialises the reader threads)."""
    import io
    import struct
    import zipfile
    import zlib
    out = {}
    with open(path, 'rb') as fh, zipfile.ZipFile(fh) as zf:
        for info in zf.infolist():
            fh.seek(info.header_offset)
            hdr = fh.read(30)
            n_name, n_extra = struct.unpack('<HH', hdr[26:30])
            fh.seek(info.header_offset + 30 + n_name + n_extra)
            raw = fh.read(info.compress_size)
            if info.compress_type == zipfile.ZIP_DEFLATED:
                raw = zlib.decompress(raw, -15, info.file_size)
            elif info.compress_type != zipfile.ZIP_STORED:
                raise ValueError("%s: unsupported zip compression %d" % (path, info.compress_type))
            bio = io.BytesIO(raw)
            major, _ = np.lib.format.read_magic(bio)
            shape, fortran, dtype = (np.lib.format.read_array_header_1_0(bio) if major == 1
                                     else np.lib.format.read_array_header_2_0(bio))
            if dtype.hasobject:
                raise ValueError("%s: object arrays are not accepted" % path)
            a = np.frombuffer(raw, dtype=dtype, offset=bio.tell(), count=int(np.prod(shape, dtype=np.int64)))
            out[info.filename[:-4] if info.filename.endswith('.npy') else info.filename] = \
                a.reshape(shape, order='F' if fortran else 'C')
    return out


def read_frame(root, stem, depth_dir=None, want_image=False, score_min=0.7, seg_dir=None) -> Frame:
    """One frame of the directory layout written by dfu3d_amd.kitti_io.write_frame."""
    pts = np.fromfile(os.path.join(root, 'velodyne', stem + '.bin'), dtype=np.float32).reshape(-1, 4)
    calib = Calibration(os.path.join(root, 'calib', stem + '.txt'))
    depth = np.load(os.path.join(depth_dir or os.path.join(root, 'depth_2'), stem + '.npy')).astype(np.float32, copy=False)
    z = _npz_arrays(os.path.join(seg_dir or os.path.join(root, 'seg_2'), stem + '.npz'))
    keep = z['scores'] > score_min                                  # vis_utils.py:218
    if keep.all():
        keep = slice(None)
    image = None
    if want_image:
        from PIL import Image
        image = np.asarray(Image.open(os.path.join(root, 'image_2', stem + '.png')).convert('RGB'), np.uint8)
    return Frame(stem, pts, calib, depth.reshape(depth.shape[0], depth.shape[1]),
                 np.ascontiguousarray(z['masks'][keep], np.uint8), z['classes'][keep].astype(np.int64),
                 z['scores'][keep].astype(np.float32), z['boxes'][keep].astype(np.float32),
                 [str(s) for s in z['thing_classes']], image, int(np.count_nonzero(z['scores'] > score_min)))


class BatchedLabeler:
    """Directory -> label files, `batch_frames` single-camera frames per engine call."""

    def __init__(self, params: Params = None, batch_frames=16, lanes=2, workers=8, device="cuda:0",
                 cap_n=1 << 17, want_points=False, reader_pool=None, max_masks=32):
        self.p = params or Params()
        self.B = int(batch_frames)
        self.lanes = int(lanes)
        self.workers = int(workers)
        self.dev = torch.device(device)
        self.cap_n = int(cap_n)
        self.want_points = bool(want_points)
        self._engines = {}
        self._copy_stream = torch.cuda.Stream(self.dev)
        self._stage = {}
        self.pool_procs = reader_pool        # dfu3d_amd.reader_pool.ReaderPool (forked before the GPU was touched) or None
        self.max_masks = int(max_masks)
        self._shm = {}
        self.seg_dir = None
        self.stats = {"frames": 0, "boxes": 0, "t_read": 0.0, "t_pack": 0.0, "t_wait": 0.0, "t_gpu": 0.0}
        self.rows_log = []          # (stem, engine rows of that frame) for the run's manifest / row gather
        self.on_frame = None        # callable(stem, rows) invoked right after the frame's label file is on disk

    # ------------------------------------------------------------------
    def _engine(self, H, W, M):
        key = (H, W, M)
        e = self._engines.get(key)
        if e is None:
            p = self.p
            if tuple(p.bounds_hw) != (H, W) or tuple(p.fov_hw) != (H, W):
                p = Params(**{**p.__dict__, "bounds_hw": (H, W), "fov_hw": (H, W)})
            per_lane = max(1, self.B // max(1, self.lanes))
            while self.B % per_lane:
                per_lane -= 1
            e = PseudoBoxEngine(p, H, W, M, self.cap_n, views_per_chunk=per_lane, dense=True,
                                cap_vox=1 << 19, pool_per_view=1 << 19, rows_per_view=256,
                                device=self.dev, lanes=self.B // per_lane)
            self._engines[key] = e
        return e

    def _staging(self, slot, B, H, W, M):
        """Pinned host buffers, two sets used alternately and kept for the whole run
        (page-locking hundreds of MB per batch costs more than the batch itself)."""
        st = self._stage.get(slot)
        if st is None or st["shape"][:3] != (B, H, W) or st["shape"][3] < M:
            st = {"shape": (B, H, W, M),
                  "pts": torch.empty((B * self.cap_n, 4), dtype=torch.float32).pin_memory(),
                  "depth": torch.zeros((B, H, W), dtype=torch.float32).pin_memory(),
                  "masks": torch.zeros(B * M * H * W, dtype=torch.uint8).pin_memory(),
                  "done": None}
            self._stage[slot] = st
        if st["done"] is not None:
            st["done"].synchronize()                   # the upload that last read these buffers
        return st

    def _instance_tensors(self, B, M, items):
        """Per-view small inputs from [(stem, calib record (48,), classes, scores, boxes, thing_classes)]."""
        calib = torch.zeros((B, 48), dtype=torch.float32)
        n_inst = torch.zeros(B, dtype=torch.int32)
        cls = torch.zeros((B, M), dtype=torch.int32)
        car = torch.zeros((B, M), dtype=torch.int32)
        rl = torch.zeros((B, M), dtype=torch.float64)
        rp = torch.zeros((B, M), dtype=torch.float64)
        box = torch.zeros((B, M, 4), dtype=torch.float32)
        score = torch.zeros((B, M), dtype=torch.float32)
        key = torch.zeros(B, dtype=torch.int64)
        for i, (stem, rec, classes, scores, boxes, names) in enumerate(items):
            m = min(len(classes), M)
            calib[i] = torch.from_numpy(np.asarray(rec, np.float32))
            n_inst[i] = m
            for j in range(m):
                nm = names[int(classes[j])]
                cls[i, j] = int(classes[j])
                car[i, j] = 1 if nm == "Car" else 0
                rl[i, j], rp[i, j] = self.p.instance_radii(nm)
            if m:
                box[i, :m] = torch.from_numpy(np.asarray(boxes[:m], np.float32))
                score[i, :m] = torch.from_numpy(np.asarray(scores[:m], np.float32))
            key[i] = int(stem) if stem.isdigit() else i
        return calib, n_inst, cls, car, rl, rp, box, score, key

    def _pack(self, frames: List[Frame], slot, pool):
        """Frames -> pinned host tensors -> device ViewBatch (async on the copy stream)."""
        B = self.B
        H, W = frames[0].depth.shape
        M = max([f.masks.shape[0] for f in frames] + [1])
        M = min(32, (M + 7) // 8 * 8)
        for f in frames:
            if f.points.shape[0] > self.cap_n:
                raise ValueError("%s: %d LiDAR points exceed cap_n=%d" % (f.stem, f.points.shape[0], self.cap_n))
            if f.depth.shape != (H, W):
                raise ValueError("%s: depth %s differs from the batch's %s" % (f.stem, f.depth.shape, (H, W)))
        n_pts = [f.points.shape[0] for f in frames] + [0] * (B - len(frames))
        off = np.zeros(B + 1, np.int64)
        off[1:] = np.cumsum(n_pts)
        st = self._staging(slot, B, H, W, M)
        pts = st["pts"][:max(int(off[-1]), 1)]
        depth = st["depth"]
        masks = st["masks"][:B * M * H * W].view(B, M, H, W)
        pts_np, depth_np, masks_np = pts.numpy(), depth.numpy(), masks.numpy()

        def fill(i):                                   # big memcpys, in parallel on the reader pool
            f = frames[i]
            m = min(f.masks.shape[0], M)
            np.copyto(pts_np[off[i]:off[i + 1]], f.points)
            np.copyto(depth_np[i], f.depth)
            if m:
                np.copyto(masks_np[i, :m], f.masks[:m])
        list(pool.map(fill, range(len(frames))))
        if len(frames) < B:
            depth[len(frames):].zero_()                # padded views: no pixels, no points, no instances
        calib, n_inst, cls, car, rl, rp, box, score, key = self._instance_tensors(
            B, M, [(f.stem, f.calib.record(), f.classes, f.scores, f.boxes, f.thing_classes) for f in frames])
        d = self.dev
        with torch.cuda.stream(self._copy_stream):
            g = lambda t: t.to(d, non_blocking=True)
            vb = ViewBatch(points=g(pts), pt_off=g(torch.from_numpy(off).to(torch.int32)),
                           view_frame=torch.arange(B, dtype=torch.int32, device=d), calib=g(calib.pin_memory()),
                           masks=g(masks), n_inst=g(n_inst.pin_memory()), inst_class=g(cls.pin_memory()),
                           inst_is_car=g(car.pin_memory()), inst_r_lidar=g(rl.pin_memory()),
                           inst_r_pseudo=g(rp.pin_memory()), inst_box=g(box.pin_memory()),
                           inst_score=g(score.pin_memory()), view_key=g(key.pin_memory()),
                           host_pt_off=off, host_view_frame=np.arange(B), depth=g(depth))
            ready = torch.cuda.Event()
            ready.record(self._copy_stream)
        st["done"] = ready
        return vb, ready, (H, W, M)

    def _write(self, frames, rows_h, label_out, npy_out, vp, skip=()):
        for i, f in enumerate(frames):
            r = rows_h[rows_h[:, 0] == i]
            if i in skip:                              # written by _label_whole_frame
                if self.on_frame is not None:
                    self.on_frame(f.stem, r)
                continue
            write_label_file(os.path.join(label_out, f.stem + '.txt'), r, NUSC_CLASSES)
            if self.on_frame is not None:
                self.on_frame(f.stem, r)
            if vp is not None:
                np.save(os.path.join(npy_out, f.stem + '.npy'), vp[i])

    def _virtual_points(self, eng, vb, frames):
        """my_loader.py:611-617 for every frame of the batch -> list of float16 arrays.
        Every chunk of the batch ran on its own lane, so lane c still holds chunk c's
        FOV list (the LiDAR rows of the file, vis_utils.py:152-154)."""
        out = [None] * len(frames)
        Vc = eng.Vc
        for c in range(self.B // Vc):
            L = eng.lanes[c]
            eng._bind(L)
            fov_n = L.n_fov.cpu().numpy()
            fov_i = L.fov_idx.view(Vc, eng.cap_n).cpu().numpy()
            n_vox, vox_pix, xyz, _ = eng.virtual_points(vb, c * Vc)
            nv = n_vox.cpu().numpy()
            for k in range(Vc):
                i = c * Vc + k
                if i >= len(frames):
                    continue
                f = frames[i]
                n = int(nv[k])
                newp = np.zeros((n, 8))
                newp[:, 0:3] = xyz[k, :n].cpu().numpy()
                if f.image is not None and n:
                    pix = vox_pix[k, :n].cpu().numpy().astype(np.int64)
                    newp[:, 4:7] = f.image.reshape(-1, 3)[pix].astype(np.int32) / 3
                newp[:, -1] = 1
                lid = f.points[fov_i[k][:fov_n[k]]]
                newl = np.zeros((lid.shape[0], 8))
                newl[:, 0:4] = lid
                newl[:, 3] *= 10
                newl[:, -1] = 2
                out[i] = np.concatenate([newl, newp], 0).astype(np.float16)
        return out

    # ------------------------------------------------------------------
    def _prepare(self, root, batch, depth_dir, pool, slot):
        torch.cuda.set_device(self.dev)
        t0 = time.perf_counter()
        frames = list(pool.map(lambda s: read_frame(root, s, depth_dir, self.want_points, seg_dir=self.seg_dir), batch))
        t1 = time.perf_counter()
        packed = self._pack(frames, slot, pool)
        self.stats["t_read"] += t1 - t0
        self.stats["t_pack"] += time.perf_counter() - t1
        return frames, packed

    # ---- reader processes: shared-memory staging, page-locked once --------------------------------
    def _shm_slot(self, slot, H, W):
        from multiprocessing import shared_memory
        from .reader_pool import ReaderPool
        st = self._shm.get(slot)
        if st is not None and (st["lay"]["H"], st["lay"]["W"]) == (H, W):
            if st["done"] is not None:
                st["done"].synchronize()
            return st
        if st is not None:
            self._release_slot(st)
        lay = ReaderPool.layout(self.B, H, W, self.cap_n, self.max_masks, self.want_points)
        shm = shared_memory.SharedMemory(create=True, size=lay["bytes"])
        buf = torch.frombuffer(shm.buf, dtype=torch.uint8)
        rc = torch.cuda.cudart().cudaHostRegister(buf.data_ptr(), lay["bytes"], 0)
        B = self.B
        esz = {torch.float32: 4, torch.uint8: 1}
        view = lambda key, shape, dt: buf[lay[key]:lay[key] + int(np.prod(shape)) * esz[dt]].view(dt).view(shape)
        st = {"lay": lay, "shm": shm, "buf": buf, "registered": (getattr(rc, "value", rc) == 0), "done": None,
              "pts": view("pts", (B, self.cap_n, 4), torch.float32), "depth": view("depth", (B, H, W), torch.float32),
              "masks": view("masks", (B, self.max_masks, H, W), torch.uint8),
              "image": view("image", (B, H, W, 3), torch.uint8) if self.want_points else None}
        st["depth"].zero_()
        self._shm[slot] = st
        return st

    def _release_slot(self, st):
        try:
            if st["registered"]:
                torch.cuda.cudart().cudaHostUnregister(st["buf"].data_ptr())
        except Exception:
            pass
        for k in ("pts", "depth", "masks", "image", "buf"):
            st[k] = None
        try:
            st["shm"].close()
            st["shm"].unlink()
        except Exception:
            pass

    def close(self):
        """Unpin and unlink the shared-memory staging (reader-process mode)."""
        for st in self._shm.values():
            self._release_slot(st)
        self._shm = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    class _Meta:
        def __init__(self, stem, n_total=0):
            self.stem, self.image, self.points, self.n_total = stem, None, None, n_total

    def _label_whole_frame(self, root, stem, label_out, depth_dir, view, npy_out=None):
        """A frame with more instances than one batch view holds (DFU3D_MAX_INST = 32): the reference loops over
        every mask (my_loader.py:547), so the frame goes through the one-frame path, which takes its instances in
        groups of 32 (penet/my_loader.py) -- same label file as `--batch-frames 0` writes.  -> engine rows with
        column 0 = `view`.  Like the batched path it needs image_2/ only when the virtual points are written, and it
        writes them to the caller's npy_out."""
        from .penet.vis_utils import load_seg_npz, save_depth_as_points
        depth = np.load(os.path.join(depth_dir or os.path.join(root, 'depth_2'), stem + '.npy')).astype(np.float32)
        _, r = save_depth_as_points(depth, stem, root, label_root=label_out, params=self.p,
                                    seg_provider=(lambda path: load_seg_npz(root, stem, self.seg_dir)),
                                    device=str(self.dev), return_rows=True, save_points=self.want_points,
                                    points_dir=npy_out)
        r = np.asarray(r, np.float64).reshape(-1, 24).copy()
        r[:, 0] = view
        return r

    def _prepare_procs(self, root, batch, depth_dir, slot):
        torch.cuda.set_device(self.dev)
        t0 = time.perf_counter()
        d0 = np.load(os.path.join(depth_dir or os.path.join(root, 'depth_2'), batch[0] + '.npy'), mmap_mode='r')
        H, W = int(d0.shape[0]), int(d0.shape[1])
        del d0
        st = self._shm_slot(slot, H, W)
        metas = self.pool_procs.read_batch(root, batch, depth_dir, self.seg_dir, self.want_points, st["shm"].name,
                                           st["lay"])
        t1 = time.perf_counter()
        B, nb = self.B, len(batch)
        if nb < B:
            st["depth"][nb:].zero_()
        M = min(self.max_masks, (max([m["m"] for m in metas] + [1]) + 7) // 8 * 8)
        n_pts = [m["n"] for m in metas] + [0] * (B - nb)
        off = np.zeros(B + 1, np.int64)
        off[1:] = np.cumsum(n_pts)
        calib, n_inst, cls, car, rl, rp, box, score, key = self._instance_tensors(
            B, M, [(m["stem"], m["calib"], m["classes"], m["scores"], m["boxes"], m["thing_classes"]) for m in metas])
        frames = []
        for i, m in enumerate(metas):
            f = BatchedLabeler._Meta(m["stem"], m.get("m_total", m["m"]))
            if self.want_points:
                f.image = st["image"][i].numpy()
                f.points = st["pts"][i, :m["n"]].numpy()
            frames.append(f)
        d = self.dev
        with torch.cuda.stream(self._copy_stream):
            g = lambda t: t.to(d, non_blocking=True)
            pts_d = torch.empty((max(int(off[-1]), 1), 4), dtype=torch.float32, device=d)
            masks_d = torch.empty((B, M, H, W), dtype=torch.uint8, device=d)
            for i, m in enumerate(metas):
                if m["n"]:
                    pts_d[off[i]:off[i + 1]].copy_(st["pts"][i, :m["n"]], non_blocking=True)
                k = min(m["m"], M)
                if k:
                    masks_d[i, :k].copy_(st["masks"][i, :k], non_blocking=True)
            vb = ViewBatch(points=pts_d, pt_off=g(torch.from_numpy(off).to(torch.int32)),
                           view_frame=torch.arange(B, dtype=torch.int32, device=d), calib=g(calib),
                           masks=masks_d, n_inst=g(n_inst), inst_class=g(cls), inst_is_car=g(car), inst_r_lidar=g(rl),
                           inst_r_pseudo=g(rp), inst_box=g(box), inst_score=g(score), view_key=g(key),
                           host_pt_off=off, host_view_frame=np.arange(B), depth=g(st["depth"]))
            ready = torch.cuda.Event()
            ready.record(self._copy_stream)
        st["done"] = ready
        self.stats["t_read"] += t1 - t0
        self.stats["t_pack"] += time.perf_counter() - t1
        return frames, (vb, ready, (H, W, M))

    def run(self, root, stems: List[str], label_out, depth_dir=None, npy_out=None, seg_dir=None):
        self.seg_dir = seg_dir
        os.makedirs(label_out, exist_ok=True)
        if self.want_points:
            npy_out = npy_out or os.path.join(root, 'velodyne_depth')
            os.makedirs(npy_out, exist_ok=True)
        batches = [stems[i:i + self.B] for i in range(0, len(stems), self.B)]
        if not batches:
            return self.stats
        pool = ThreadPoolExecutor(self.workers)       # file readers
        prep = ThreadPoolExecutor(1)                  # reads + packs + uploads the NEXT batch
        writers = ThreadPoolExecutor(2)
        pending = []
        prepare = (lambda b_, slot_: self._prepare_procs(root, b_, depth_dir, slot_)) if self.pool_procs is not None \
            else (lambda b_, slot_: self._prepare(root, b_, depth_dir, pool, slot_))
        fut = prep.submit(prepare, batches[0], 0)
        for bi in range(len(batches)):
            t0 = time.perf_counter()
            frames, (vb, ready, (H, W, M)) = fut.result()
            t1 = time.perf_counter()
            self.stats["t_wait"] += t1 - t0
            if bi + 1 < len(batches):
                fut = prep.submit(prepare, batches[bi + 1], (bi + 1) & 1)
            eng = self._engine(H, W, M)
            cur = torch.cuda.current_stream(self.dev)
            cur.wait_event(ready)
            for t in vb.__dict__.values():             # uploaded on the copy stream, used on this one
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(cur)
            rows, status = eng.run(vb)
            if status:
                from .stages import status_message
                raise RuntimeError("device status: " + status_message(status))
            rows_h = rows.cpu().numpy()
            self.stats["t_gpu"] += time.perf_counter() - t1
            vp = self._virtual_points(eng, vb, frames) if self.want_points else None
            whole = [i for i, f in enumerate(frames) if getattr(f, "n_total", 0) > self.max_masks]
            for i in whole:                            # never a silent cut: every instance of the frame is labelled
                import sys
                print("[dfu3d] %s: %d instances, more than the %d of a batch view: labelled through the one-frame path"
                      % (frames[i].stem, frames[i].n_total, self.max_masks), file=sys.stderr)
                rows_h = np.concatenate([rows_h[rows_h[:, 0] != i],
                                         self._label_whole_frame(root, frames[i].stem, label_out, depth_dir, i, npy_out)], 0)
                self.stats["frames_over_32_instances"] = self.stats.get("frames_over_32_instances", 0) + 1
            pending.append(writers.submit(self._write, frames, rows_h, label_out, npy_out, vp, set(whole)))
            self.stats["frames"] += len(frames)
            self.stats["boxes"] += int(rows_h.shape[0])
            for i, f in enumerate(frames):
                self.rows_log.append((f.stem, rows_h[rows_h[:, 0] == i]))
        for p_ in pending:
            p_.result()
        for ex in (pool, prep, writers):
            ex.shutdown()
        return self.stats
