"""Dev tool (GPU): time dfu3d_range_cluster / dfu3d_lshape_fit per segment on a
synthetic batch to see which instances dominate a launch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dfu3d_amd import synth, stages as st
from dfu3d_amd.engine import PseudoBoxEngine
from dfu3d_amd.params import Params

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = "cuda:0"
p = Params()
scenes = [synth.make_scene(f, dense=True, device=dev, k_min=30, k_max=40) for f in range(frames)]
b = synth.to_view_batch(scenes, p, dev)
eng = PseudoBoxEngine(p, 900, 1600, 8, 34720, views_per_chunk=frames * 6, pool_per_view=1 << 17)
rows, status = eng.run(b)
torch.cuda.synchronize()
S = eng.Vc * eng.M
cnt = eng.cnt_all.clone()
order = torch.argsort(cnt, descending=True)[:12].tolist()
print("total pts", int(cnt.sum()), "segments>0", int((cnt > 0).sum()))
def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
full = timeit(lambda: st.range_cluster(eng.px, eng.py, eng.base_a, cnt, S, p.R0, p.Rd, eng.label, eng.pool_cap))
print("range_cluster all segments: %.3f ms" % full)
for s in order:
    one = torch.zeros_like(cnt); one[s] = cnt[s]
    t = timeit(lambda: st.range_cluster(eng.px, eng.py, eng.base_a, one, S, p.R0, p.Rd, eng.label, eng.pool_cap))
    lab = eng.label[int(eng.base_a[s]):int(eng.base_a[s]) + int(cnt[s])]
    ncl = int((lab == torch.arange(lab.numel(), device=dev)).sum())
    print("seg %4d n=%6d clusters=%3d  %.3f ms" % (s, int(cnt[s]), ncl, t))
none = torch.zeros_like(cnt)
print("empty launch: %.3f ms" % timeit(lambda: st.range_cluster(eng.px, eng.py, eng.base_a, none, S, p.R0, p.Rd, eng.label, eng.pool_cap)))
small = cnt.clone(); small[cnt > 4096] = 0
print("only n<=4096: %.3f ms" % timeit(lambda: st.range_cluster(eng.px, eng.py, eng.base_a, small, S, p.R0, p.Rd, eng.label, eng.pool_cap)))
large = cnt.clone(); large[cnt <= 4096] = 0
print("only n>4096 (%d segs): %.3f ms" % (int((large > 0).sum()), timeit(lambda: st.range_cluster(eng.px, eng.py, eng.base_a, large, S, p.R0, p.Rd, eng.label, eng.pool_cap))))
for k in (2, 4, 8, 16):
    sel = torch.zeros_like(cnt); sel[order[:k]] = cnt[order[:k]]
    print("top-%d together: %.3f ms" % (k, timeit(lambda: st.range_cluster(eng.px, eng.py, eng.base_a, sel, S, p.R0, p.Rd, eng.label, eng.pool_cap))))
print("segment ids of top:", order)
allorder = torch.argsort(cnt, descending=True).tolist()
for s_ in allorder[12:30]:
    if int(cnt[s_]) <= 3000: break
    one = torch.zeros_like(cnt); one[s_] = cnt[s_]
    t = timeit(lambda: st.range_cluster(eng.px, eng.py, eng.base_a, one, S, p.R0, p.Rd, eng.label, eng.pool_cap))
    b0 = int(eng.base_a[s_]); n_ = int(cnt[s_])
    lab = eng.label[b0:b0 + n_]
    roots, sizes = torch.unique(lab, return_counts=True)
    xs = eng.px[b0:b0+n_]; ys = eng.py[b0:b0+n_]
    print("seg %4d n=%6d clusters=%3d top sizes %s extent %.1f x %.1f  %.3f ms" % (s_, n_, roots.numel(), sorted(sizes.tolist())[::-1][:5], float(xs.max()-xs.min()), float(ys.max()-ys.min()), t))
