"""Disk-inclusive throughput of the batched directory labeler.

Writes SAMPLES x 6 single-camera KITTI frames (900x1600, the layout of tools/nuscenes2kitti.py:246-285
+ seg_2 / depth_2) under a scratch directory, then times dfu3d_amd.pipeline.BatchedLabeler over them:
file reads -> pinned staging -> async upload -> multi-view engine -> label files.  This is NOT the number
bench.py reports (bench.py starts with the inputs resident in HBM); DESIGN.md quotes it separately.

usage: python tools/bench_directory.py [samples=8] [batch_frames=24] [streams=2] [workers=12] [scratch=/dev/shm/dfu3d_dir]
"""
import json
import os
import shutil
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
procs = int(sys.argv[7]) if len(sys.argv) > 7 else 0
rpool = None
if procs > 0:                                  # before anything touches the GPU
    from dfu3d_amd.reader_pool import ReaderPool
    rpool = ReaderPool(procs)
import numpy as np
import torch

from dfu3d_amd import kitti_io, synth
from dfu3d_amd.params import NUSC_CLASSES
from dfu3d_amd.pipeline import BatchedLabeler, read_frame

samples = int(sys.argv[1]) if len(sys.argv) > 1 else 8
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 24
streams = int(sys.argv[3]) if len(sys.argv) > 3 else 2
workers = int(sys.argv[4]) if len(sys.argv) > 4 else 12
root = sys.argv[5] if len(sys.argv) > 5 else "/dev/shm/dfu3d_dir"
compress = (sys.argv[6] != "raw") if len(sys.argv) > 6 else True
H, W, M, CAMS = 900, 1600, 8, 6

shutil.rmtree(root, ignore_errors=True)
t0 = time.perf_counter()
idx = 0
img = np.zeros((H, W, 3), np.uint8)
for s in range(samples):
    sc = synth.make_scene(s, H=H, W=W, M=M, cams=CAMS, dense=True, device="cuda:0")
    pts = sc.points.cpu().numpy()
    for c in range(CAMS):
        n = int(sc.n_inst[c])
        kitti_io.write_frame(root, idx, pts, sc.calibs[c], img, sc.masks[c][:n].cpu().numpy(),
                             sc.inst_class[c][:n].cpu().numpy(), np.full(n, 0.9, np.float32),
                             sc.inst_box[c][:n].cpu().numpy(), NUSC_CLASSES, sc.depth[c].cpu().numpy(), compress=compress)
        idx += 1
t_write = time.perf_counter() - t0
stems = ["%06d" % i for i in range(idx)]
bytes_in = sum(os.path.getsize(os.path.join(root, d, f)) for d in ("velodyne", "calib", "seg_2", "depth_2")
               for f in os.listdir(os.path.join(root, d)))

# host-side read cost of ONE frame on one thread (np.fromfile + calib parse + npy + npz inflate)
t0 = time.perf_counter()
for s_ in stems[:6]:
    read_frame(root, s_)
t_read1 = (time.perf_counter() - t0) / 6

out = {}
lab = BatchedLabeler(batch_frames=batch, lanes=streams, workers=workers, cap_n=1 << 16, reader_pool=rpool)
for rep in range(5):                                   # first pass builds the engine, pins the staging buffers
    for k in lab.stats:
        lab.stats[k] = 0
    shutil.rmtree(os.path.join(root, "label_2"), ignore_errors=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = lab.run(root, stems, os.path.join(root, "label_2"))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["pass%d" % rep] = {"seconds": round(dt, 3), "camera_frames_per_s": round(idx / dt, 1),
                           "samples_per_s": round(idx / dt / CAMS, 2), "boxes": st["boxes"],
                           **{k: round(v, 3) for k, v in st.items() if k.startswith("t_")}}
# the same engine call with nothing else running on the host (no reader threads)
from concurrent.futures import ThreadPoolExecutor
pool = ThreadPoolExecutor(workers)
lab.pool_procs = None
frames, (vb, ready, (h_, w_, m_)) = lab._prepare(root, stems[:batch], None, pool, 0)
ready.synchronize()
eng = lab._engine(h_, w_, m_)
for _ in range(3):
    rows, status = eng.run(vb); rows.cpu()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    rows, status = eng.run(vb); rows.cpu()
torch.cuda.synchronize()
out["engine_only_ms_per_batch"] = round((time.perf_counter() - t0) * 100, 2)
out.update({"camera_frames": idx, "input_MB": round(bytes_in / 1e6, 1), "read_one_frame_ms_one_thread": round(t_read1 * 1e3, 1),
            "write_dataset_s": round(t_write, 1), "batch_frames": batch, "streams": streams, "workers": workers,
            "cpus": os.cpu_count(), "scratch": root, "masks": "deflated npz" if compress else "stored npz"})
out["reader_procs"] = procs
print(json.dumps(out))
lab.close()
if rpool is not None:
    rpool.close()
shutil.rmtree(root, ignore_errors=True)
