"""Dev tool (GPU): the pool-overflow case of tests/test_gpu_edge_cases.py stage by stage with a sync after every stage."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dfu3d_amd import synth, stages as st
from dfu3d_amd.engine import PseudoBoxEngine
from dfu3d_amd.params import Params
H, W, M, cams = 180, 320, 5, 2
p = Params(bounds_hw=(H, W), fov_hw=(H, W))
s = synth.make_scene(90, H=H, W=W, M=M, cams=cams, dense=True, k_min=14, k_max=18)
b = synth.to_view_batch([s], p, "cuda:0", dense=True)
n = s.points.shape[0]
eng = PseudoBoxEngine(p, H, W, M, n, views_per_chunk=cams, dense=True, cap_vox=1 << 16, pool_per_view=int(sys.argv[1]) if len(sys.argv) > 1 else 64)
orig = eng._run
def run(name, fn, *a, **k):
    print("stage", name, flush=True)
    r = fn(*a, **k)
    torch.cuda.synchronize()
    print("   ok", flush=True)
    return r
eng._run = run
eng.timing = True      # phase by phase
eng._events = []
def phased(prefix, fn, phases, *a, **kw):
    for tag, bit in phases:
        run(prefix + tag, fn, *a, phases=bit, **kw)
eng._phased = phased
rows, status = eng.run(b)
print("status", status, rows.shape, "cursor", int(eng.pool_cursor), "cnt_a", eng.cnt_a.tolist(), "cnt_b", eng.cnt_b.tolist())
