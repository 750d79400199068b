"""Dev tool (GPU box): per-phase cycles of the radius filter's two kernels on the bench workload (variant build rf_timing:
clock64 of thread 0 / lane 0 at the phase marks, one record per workgroup / wave).

    python tools/rf_timing.py [frames=64]
"""
import ctypes
import os
import sys

os.environ["DFU3D_LIB_VARIANT"] = "rf_timing"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from dfu3d_amd import _lib, synth
from dfu3d_amd.engine import PseudoBoxEngine
from dfu3d_amd.params import Params

dev = torch.device("cuda", 0)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 64
params = Params()
H, W, M, CAMS, N_PTS = 900, 1600, 8, 6, 34720
scenes = [synth.make_scene(f, H=H, W=W, M=M, cams=CAMS, dense=True, device=dev, k_min=30, k_max=40) for f in range(frames)]
batch = synth.to_view_batch(scenes, params, dev, dense=True, frame_ids=list(range(frames)))
batch.pack_masks()
del scenes
eng = PseudoBoxEngine(params, H, W, M, N_PTS, views_per_chunk=frames * CAMS, dense=True, cap_vox=1 << 18,
                      pool_per_view=1 << 17, device=dev, lanes=1, chain=True)
L = _lib.lib()
F = 12
A = np.zeros(32768 * F, np.uint64)
B = np.zeros(16384 * F, np.uint64)
pa = A.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong))
pb = B.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong))
for _ in range(3):
    eng.run(batch)
torch.cuda.synchronize()
L.dfu3d_debug_rf_timing(pa, pb, 1)
eng.run(batch)
torch.cuda.synchronize()
L.dfu3d_debug_rf_timing(pa, pb, 0)
A = A.reshape(-1, F).astype(np.int64)
B = B.reshape(-1, F).astype(np.int64)
A = A[A[:, 5] > 0]
B = B[B[:, 3] > 0]
print("phase A: %d working workgroups, %.1f listed points each (most %d)" % (len(A), A[:, 6].mean(), A[:, 6].max()))
for i, nm in enumerate(["loads issued", "eight chunks", "box + U lists + barrier", "pairwise", "queue"]):
    print("  %-28s mean %8.0f   median %8.0f   p99 %8.0f   max %8d cycles" % (nm, A[:, i].mean(), np.median(A[:, i]), np.percentile(A[:, i], 99), A[:, i].max()))
life = (A[:, 9] - A[:, 8]) / 100.0
span = (A[:, 9].max() - A[:, 8].min()) / 100.0
print("  lifetime: mean %.2f us, median %.2f, p99 %.2f, longest %.2f (100 MHz clock); first start -> last end %.1f us; mean concurrency %.1f workgroups"
      % (life.mean(), np.median(life), np.percentile(life, 99), life.max(), span, life.sum() / span))
print("  shader clock while the workgroups ran: %.2f GHz" % (A[:, 0:5].sum() / max((A[:, 9] - A[:, 8]).sum(), 1) / 10.0))
# start times: how fast does the grid start?
st = np.sort(A[:, 8] - A[:, 8].min()) / 100.0
print("  workgroup starts: 10%% by %.1f us, 50%% by %.1f, 90%% by %.1f, last %.1f" % (st[len(st) // 10], st[len(st) // 2], st[len(st) * 9 // 10], st[-1]))
q = max(int(B[:, 3].sum()), 1)
print("phase B: %d queries (%d kept) on %d waves, U list %.0f entries on average (longest %d), %.1f boxes, %.2f candidate ranges on average (most %d)"
      % (q, B[:, 7].sum(), len(B), B[:, 4].sum() / q, B[:, 4].max(), B[:, 6].sum() / q, B[:, 5].sum() / q, B[:, 5].max()))
for i, nm in enumerate(["set-up (queue, point, tables)", "U list", "boxes + candidate ranges"]):
    print("  %-30s mean %8.0f   median %8.0f   p99 %8.0f   max %8d cycles / wave" % (nm, B[:, i].mean(), np.median(B[:, i]), np.percentile(B[:, i], 99), B[:, i].max()))
tot = B[:, 0:3].sum(1)
k = int(np.argmax(tot))
print("  longest wave: %d cycles: U list %d entries, %d candidate ranges, segment of %d points" % (tot[k], B[k, 4], B[k, 5], B[k, 8]))
order = np.argsort(-tot)[:8]
for k in order:
    print("    %8d cycles  setup %7d  ulist %7d  boxes %7d   nu %5d cand %3d  n %6d" % (tot[k], B[k, 0], B[k, 1], B[k, 2], B[k, 4], B[k, 5], B[k, 8]))
