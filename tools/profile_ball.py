"""Dev tool (GPU): per-segment timing of dfu3d_ballquery_fuse (state snapshotted before the fuse)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dfu3d_amd import synth, stages as st
from dfu3d_amd.engine import PseudoBoxEngine
from dfu3d_amd.params import Params
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = "cuda:0"; p = Params()
scenes = [synth.make_scene(f, dense=True, device=dev, k_min=30, k_max=40) for f in range(frames)]
b = synth.to_view_batch(scenes, p, dev)
eng = PseudoBoxEngine(p, 900, 1600, 8, 34720, views_per_chunk=frames * 6, pool_per_view=1 << 17)
snap = {}
orig = st.ballquery_fuse
def hook(px, py, pz, base_a, cnt_a, base_b, cnt_b, C, S, cap, tile_off, flags, masked=False):
    snap.update(px=px.clone(), py=py.clone(), pz=pz.clone(), base_a=base_a.clone(), cnt_a=cnt_a.clone(), base_b=base_b.clone(),
                cnt_b=cnt_b.clone(), flags=flags.clone(), masked=masked)
    return orig(px, py, pz, base_a, cnt_a, base_b, cnt_b, C, S, cap, tile_off, flags, masked=masked)
st.ballquery_fuse = hook
eng.run(b); torch.cuda.synchronize()
st.ballquery_fuse = orig
S = eng.Vc * eng.M
def run(cb):
    px, py, pz = snap["px"].clone(), snap["py"].clone(), snap["pz"].clone()
    bb, cb2 = snap["base_b"].clone(), cb.clone()
    fl = snap["flags"].clone()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    orig(px, py, pz, snap["base_a"], snap["cnt_a"], bb, cb2, p.fuse_C, S, eng.pool_cap, eng.tile_off, fl, masked=snap["masked"])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3, cb2
ca, cb = snap["cnt_a"], snap["cnt_b"]
run(cb)
t = min(run(cb)[0] for _ in range(4)); out = run(cb)[1]
print("ball all: %.3f ms; pseudo in %d, kept %d; lidar %d" % (t, int(cb.sum()), int(out.sum()), int(ca.sum())))
work = (ca.long() * cb.long())
for s_ in torch.argsort(work, descending=True)[:10].tolist():
    one = torch.zeros_like(cb); one[s_] = cb[s_]
    run(one); t, o2 = run(one)
    print("seg %4d na=%5d nq=%6d kept=%6d  %.3f ms" % (s_, int(ca[s_]), int(cb[s_]), int(o2[s_]), t))
z = torch.zeros_like(cb); print("empty: %.3f ms" % run(z)[0])
