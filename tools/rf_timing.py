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
P = np.zeros(32768 * F, np.uint64)
B = np.zeros(16384 * F, np.uint64)
ptr = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong))
for _ in range(3):
    eng.run(batch)
torch.cuda.synchronize()
L.dfu3d_debug_rf_timing(ptr(A), ptr(P), ptr(B), 1)
eng.run(batch)
torch.cuda.synchronize()
L.dfu3d_debug_rf_timing(ptr(A), ptr(P), ptr(B), 0)
A = A.reshape(-1, F).astype(np.int64)
P = P.reshape(-1, F).astype(np.int64)
B = B.reshape(-1, F).astype(np.int64)
A = A[A[:, 5] > 0]
P = P[P[:, 5] > 0]
B = B[B[:, 3] > 0]
print("phase A: %d ranges, %.1f listed points each (most %d)" % (len(A), A[:, 6].mean(), A[:, 6].max()))
for i, nm in enumerate(["loads issued", "eight chunks", "box + record"]):
    print("  %-30s mean %8.0f   median %8.0f   p99 %8.0f   max %8d cycles" % (nm, A[:, i].mean(), np.median(A[:, i]), np.percentile(A[:, i], 99), A[:, i].max()))
life = (A[:, 9] - A[:, 8]) / 100.0
span = (A[:, 9].max() - A[:, 8].min()) / 100.0
print("  lifetime: mean %.2f us, median %.2f, p99 %.2f, longest %.2f (100 MHz clock); first start -> last end %.1f us; mean concurrency %.1f waves"
      % (life.mean(), np.median(life), np.percentile(life, 99), life.max(), span, life.sum() / span))
st = np.sort(A[:, 8] - A[:, 8].min()) / 100.0
print("  wave starts: 10%% by %.1f us, 50%% by %.1f, 90%% by %.1f, last %.1f" % (st[len(st) // 10], st[len(st) // 2], st[len(st) * 9 // 10], st[-1]))
print("phase A': %d ranges with listed points, %.1f queries and %.1f partners each (most %d / %d)"
      % (len(P), P[:, 6].mean(), P[:, 7].mean(), P[:, 6].max(), P[:, 7].max()))
for i, nm in enumerate(["records, slots", "box, pairwise, flags, queue"]):
    print("  %-30s mean %8.0f   median %8.0f   p99 %8.0f   max %8d cycles" % (nm, P[:, i].mean(), np.median(P[:, i]), np.percentile(P[:, i], 99), P[:, i].max()))
q = max(int(B[:, 3].sum()), 1)
print("phase B: %d queries (%d kept, %d of short segments read whole) on %d waves, %.1f ranges, %.2f candidate ranges on average (most %d)"
      % (q, B[:, 7].sum(), B[:, 4].sum(), len(B), B[:, 6].sum() / q, B[:, 5].sum() / q, B[:, 5].max()))
nd = B[B[:, 4] == 0]
print("  long segments: %d queries, %d settled by the ranges next door, %d kept in the end" % (len(nd), nd[:, 9].sum(), nd[:, 7].sum()))
for i, nm in enumerate(["set-up (queue, point, tables)", "whole segment / slots next door", "all ranges, candidates, flag"]):
    print("  %-32s mean %8.0f   median %8.0f   p99 %8.0f   max %8d cycles / wave" % (nm, B[:, i].mean(), np.median(B[:, i]), np.percentile(B[:, i], 99), B[:, i].max()))
tot = B[:, 0:3].sum(1)
for k in np.argsort(-tot)[:8]:
    print("    %8d cycles  setup %7d  part1 %7d  part2 %7d   queries %d cand %3d  n %6d" % (tot[k], B[k, 0], B[k, 1], B[k, 2], B[k, 3], B[k, 5], B[k, 8]))
