"""Dev tool (GPU box): per-phase cycles of k_range_cluster_grid on the bench workload (variant build grid_timing)."""
import os, sys, ctypes
os.environ["DFU3D_LIB_VARIANT"] = "grid_timing"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dfu3d_amd import synth, _lib
from dfu3d_amd.engine import PseudoBoxEngine
from dfu3d_amd.params import Params
dev = torch.device("cuda", 0)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 16
params = Params()
H, W, M, CAMS, N_PTS = 900, 1600, 8, 6, 34720
scenes = [synth.make_scene(f, H=H, W=W, M=M, cams=CAMS, dense=True, device=dev, k_min=30, k_max=40) for f in range(frames)]
batch = synth.to_view_batch(scenes, params, dev, dense=True, frame_ids=list(range(frames)))
batch.pack_masks()
del scenes
eng = PseudoBoxEngine(params, H, W, M, N_PTS, views_per_chunk=frames * CAMS, dense=True, cap_vox=1 << 18,
                      pool_per_view=1 << 17, device=dev, lanes=1, chain=True)
L = _lib.lib()
out = (ctypes.c_ulonglong * 16)()
eng.run(batch); torch.cuda.synchronize()
L.dfu3d_debug_grid_timing(out, 1)
L.dfu3d_debug_grid_max(out, 1)
eng.run(batch); torch.cuda.synchronize()
L.dfu3d_debug_grid_timing(out, 1)
v = list(out)
L.dfu3d_debug_grid_max(out, 1)
mx = list(out)
names = ["geometry", "sort", "link touching", "link gap", "labels"]
wg = max(v[8], 1)
print("workgroups", v[8], "points", v[9], "cells", v[10], "wave-cooperative pair tests", v[11])
for i, nm in enumerate(names):
    print("%-14s %10.1f cycles/WG   max %9d cycles (n = %d)" % (nm, v[i] / wg, mx[i] >> 24, mx[i] & 0xFFFFFF))
print("slowest workgroup: %d cycles, n = %d;   largest instance: n = %d, %d cycles"
      % (v[12] >> 24, v[12] & 0xFFFFFF, v[13] >> 32, v[13] & 0xFFFFFFFF))
