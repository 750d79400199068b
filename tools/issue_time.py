"""Dev tool (GPU): how long the host needs to ENQUEUE one step (all chunks) versus how long the GPU needs to run it."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dfu3d_amd import synth
from dfu3d_amd.engine import PseudoBoxEngine
from dfu3d_amd.params import Params
frames, chunk, lanes = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (64, 16, 4)))
p = Params(); dev = "cuda:0"
scenes = [synth.make_scene(f, dense=True, device=dev, k_min=30, k_max=40) for f in range(frames)]
b = synth.to_view_batch(scenes, p, dev)
eng = PseudoBoxEngine(p, 900, 1600, 8, 34720, views_per_chunk=chunk * 6, pool_per_view=1 << 17, lanes=lanes)
for _ in range(3): eng.run(b)
torch.cuda.synchronize()
ti, tt = [], []
for _ in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.run(b, sync=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ti.append(t1 - t0); tt.append(t2 - t0)
print("frames %d chunk %d lanes %d: enqueue %.2f ms, until GPU idle %.2f ms" % (frames, chunk, lanes, min(ti) * 1e3, min(tt) * 1e3))
