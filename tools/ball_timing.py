"""Dev tool (GPU box): per-workgroup records of the two k_ball_flags builds on the bench workload (variant build
ball_timing): when each workgroup started, how long it ran (100 MHz wall clock), its instance's sizes and the longest
chain walk of its lanes.

    python tools/ball_timing.py [frames=64]
"""
import os, sys, ctypes
os.environ["DFU3D_LIB_VARIANT"] = "ball_timing"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dfu3d_amd import synth, _lib
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
L.dfu3d_debug_ball_timing.restype = ctypes.c_int
NREC = 1 << 17
buf = (ctypes.c_ulonglong * (2 * NREC * 4))()
eng.run(batch); torch.cuda.synchronize()
L.dfu3d_debug_ball_timing(buf, 1)
eng.run(batch); torch.cuda.synchronize()
L.dfu3d_debug_ball_timing(buf, 0)
a = np.frombuffer(buf, dtype=np.uint64).reshape(2, NREC, 4)
for b, name in ((0, "small build (256 threads, <= 1024 LiDAR points)"), (1, "big build (1024 threads)")):
    r = a[b]
    used = r[:, 0] != 0
    start = (r[used, 0] >> np.uint64(32)).astype(np.int64)
    dur = (r[used, 0] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    na = (r[used, 1] >> np.uint64(40)).astype(np.int64)
    nq = ((r[used, 1] >> np.uint64(20)) & np.uint64(0xFFFFF)).astype(np.int64)
    walk = (r[used, 1] & np.uint64(0xFFFFF)).astype(np.int64)
    if len(dur) == 0:
        print(name, ": no workgroups"); continue
    t0 = start.min()
    end = ((start - t0) & 0xFFFFFFFF) + dur
    print("%s: %d workgroups with a tile; span of the kernel %.1f us; duration of a workgroup: mean %.1f, median %.1f, p99 %.1f, max %.1f us"
          % (name, len(dur), end.max() / 100.0, dur.mean() / 100.0, np.median(dur) / 100.0, np.percentile(dur, 99) / 100.0, dur.max() / 100.0))
    print("   sum of the durations / (256 CUs x workgroups per CU): %.1f us at 8 per CU, %.1f at 2" % (dur.sum() / 100.0 / 2048, dur.sum() / 100.0 / 512))
    for i in np.argsort(-dur)[:8]:
        print("   %7.1f us  started at %7.1f  na %5d  nq %6d  most exact tests of a lane %4d (that lane walked %4d nodes)" % (dur[i] / 100.0, ((start[i] - t0) & 0xFFFFFFFF) / 100.0, na[i], nq[i], walk[i] >> 12, walk[i] & 0xFFF))
    # start-time profile: how many workgroups had started by ...
    rel = ((start - t0) & 0xFFFFFFFF) / 100.0
    ph = r[used, 2]
    p1 = (ph >> np.uint64(40)).astype(np.int64) / 100.0; p2 = ((ph >> np.uint64(20)) & np.uint64(0xFFFFF)).astype(np.int64) / 100.0
    p3 = (ph & np.uint64(0xFFFFF)).astype(np.int64) / 100.0
    print("   thread 0: tile facts + query point loaded after %.1f us (median), table built after another %.1f, queries done after another %.1f; rest %.1f"
          % (np.median(p1), np.median(p2), np.median(p3), np.median(dur / 100.0 - p1 - p2 - p3)))
    print("   most exact tests of a lane per workgroup: mean %.1f, median %.1f, p99 %.1f" % ((walk >> 12).mean(), np.median(walk >> 12), np.percentile(walk >> 12, 99)))
    print("   workgroups started by 25/50/75/100 %% of the span: %s" % [int((rel <= q * end.max() / 100.0).sum()) for q in (0.25, 0.5, 0.75, 1.0)])
