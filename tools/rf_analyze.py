"""Dev tool (GPU): who are the points of the bench pool that have no list neighbour (position +-1) within the radius?
For each such point: the list distance to the nearest point of its segment within the radius (searched +-W positions)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dfu3d_amd import stages as st, synth
from dfu3d_amd.engine import PseudoBoxEngine
from dfu3d_amd.params import Params
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = "cuda:0"; p = Params()
scenes = [synth.make_scene(f, dense=True, device=dev, k_min=30, k_max=40) for f in range(frames)]
b = synth.to_view_batch(scenes, p, dev); b.pack_masks()
eng = PseudoBoxEngine(p, 900, 1600, 8, 34720, views_per_chunk=frames * 6)
snaps = []; orig = st.radius_filter
def hook(*a, **k):
    if not snaps: snaps.append([x.clone() if isinstance(x, torch.Tensor) else x for x in a])
    return orig(*a, **k)
st.radius_filter = hook; eng.run(b); torch.cuda.synchronize(); st.radius_filter = orig
px, py, pz, base, cnt, rad = snaps[0][:6]
S2 = base.numel()
n_used = int((base + cnt).max())
X = torch.stack([px, py, pz], 1)[:n_used]
seg = torch.full((n_used,), -1, dtype=torch.int64, device=dev)
order = torch.argsort(base)
bs, cs = base[order], cnt[order].long()
seg_of = torch.repeat_interleave(order, cs)
pos = torch.cat([torch.arange(int(b0), int(b0) + int(c0), device=dev) for b0, c0 in zip(bs.tolist(), cs.tolist()) if c0 > 0])
seg[pos] = seg_of
r = rad[seg.clamp(min=0)]
active = (seg >= 0) & (r > 0)
def within(d):
    j = torch.arange(n_used, device=dev) + d
    ok = (j >= 0) & (j < n_used)
    jc = j.clamp(0, n_used - 1)
    dd = ((X - X[jc]) ** 2).sum(1)
    return ok & (seg[jc] == seg) & (dd < r * r)
coh = within(-1) | within(1)
U = active & ~coh
print("points %d active %d; incoherent (no +-1 list neighbour within r): %d (%.2f%%)" % (n_used, int(active.sum()), int(U.sum()), 100.0 * int(U.sum()) / int(active.sum())))
is_lidar = seg < S2 // 2
print("  of them LiDAR lists: %d of %d LiDAR points; pseudo lists: %d of %d" % (int((U & is_lidar).sum()), int((active & is_lidar).sum()), int((U & ~is_lidar).sum()), int((active & ~is_lidar).sum())))
for rr in (3.0, 0.6):
    m = active & (r == rr)
    print("  radius %.1f: %d points, %d incoherent" % (rr, int(m.sum()), int((U & m).sum())))
near = torch.full((n_used,), 10 ** 9, dtype=torch.int64, device=dev)
W = 512
for d in list(range(2, W + 1)):
    for sgn in (-1, 1):
        hit = U & within(sgn * d) & (near > d)
        near[hit] = d
for nm, m in (("pseudo", U & ~is_lidar), ("lidar", U & is_lidar)):
    nn = near[m].cpu().numpy()
    tot = len(nn)
    if tot == 0: continue
    edges = [2, 3, 4, 8, 16, 32, 64, 128, 256, 512]
    msg = ["%s: %d incoherent;" % (nm, tot)]
    prev = 1
    for e in edges:
        msg.append("<=%d: %.1f%%" % (e, 100.0 * ((nn <= e).sum()) / tot))
    msg.append("none within %d: %.1f%%" % (W, 100.0 * (nn > W).sum() / tot))
    print("  " + " ".join(msg))
