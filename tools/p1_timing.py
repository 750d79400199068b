"""Dev tool (GPU box): per-phase cycles of k_bp_bin on the bench workload (variant build `timing`: csrc/dbg.hpp): cycles of thread 0
of every workgroup between the phase marks, averaged over the workgroups."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dfu3d_amd import synth, _lib
from dfu3d_amd.engine import PseudoBoxEngine
from dfu3d_amd.params import Params
dev = torch.device("cuda", 0)
L = _lib.load_variant("timing")
_lib._LIB = L          # this process drives the timing build through the ordinary engine
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 16
params = Params()
H, W, M, CAMS, N_PTS = 900, 1600, 8, 6, 34720
scenes = [synth.make_scene(f, H=H, W=W, M=M, cams=CAMS, dense=True, device=dev, k_min=30, k_max=40) for f in range(frames)]
batch = synth.to_view_batch(scenes, params, dev, dense=True, frame_ids=list(range(frames)))
batch.pack_masks()
del scenes
eng = PseudoBoxEngine(params, H, W, M, N_PTS, views_per_chunk=frames * CAMS, dense=True, cap_vox=1 << 18,
                      pool_per_view=1 << 17, device=dev, lanes=1, chain=True)
L.dfu3d_debug_timing_pixel.restype = ctypes.c_int
out = (ctypes.c_ulonglong * 32)()
eng.run(batch); torch.cuda.synchronize()
L.dfu3d_debug_timing_pixel(out, 1)
eng.run(batch); torch.cuda.synchronize()
L.dfu3d_debug_timing_pixel(out, 1)
v = list(out)
wg = max(v[8], 1)
names = ["set-up + barrier", "loads, classification, exact keys", "origin reduction + barrier", "run merging + LDS atomics",
         "barrier", "flush (global atomics issued)", "barrier", "bit-map flush"]
tot = sum(v[:8])
print("k_bp_bin: %d workgroups, %.0f cycles per workgroup (thread 0)" % (v[8], tot / wg))
for i, nm in enumerate(names):
    print("  %-36s %9.0f cycles  %5.1f %%" % (nm, v[i] / wg, 100.0 * v[i] / max(tot, 1)))
