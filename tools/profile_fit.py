"""Dev tool (GPU): per-segment timing of dfu3d_lshape_fit / ballquery on a synthetic batch."""
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
def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
rows_b = torch.zeros(eng.cap_rows * st.ROW_DOUBLES, dtype=torch.float64, device=dev)
nr = torch.zeros(1, dtype=torch.int32, device=dev); stt = torch.zeros(1, dtype=torch.int32, device=dev)
def fit(c):
    nr.zero_()
    st.lshape_fit(eng.px, eng.py, eng.pz, eng.label, eng.base_a, c, S, eng.M, b.calib, b.inst_class.reshape(-1),
                  b.inst_is_car.reshape(-1), b.inst_box.reshape(-1), b.inst_score.reshape(-1), eng.n_theta, eng.dtheta,
                  p.car_aspect_max, eng.sx, eng.sy, eng.sroot, eng.cap_rows, rows_b, nr, stt, eng.pool_cap)
print("lshape_fit all: %.3f ms" % timeit(lambda: fit(cnt)))
# cluster-size histogram over the whole batch
sizes_all = []
cn = cnt.cpu().numpy(); ba = eng.base_a.cpu().numpy(); lab_all = eng.label.cpu().numpy()
nclus = []
for s_ in range(S):
    if cn[s_] == 0: continue
    l = lab_all[ba[s_]:ba[s_] + cn[s_]]
    u, c = np.unique(l, return_counts=True)
    sizes_all += c.tolist(); nclus.append(len(u))
sizes_all = np.array(sizes_all)
print("segments %d (nonempty %d), points %d, clusters %d, clusters/segment mean %.2f max %d" % (S, (cn > 0).sum(), cn.sum(), len(sizes_all), np.mean(nclus), max(nclus)))
print("segment size pct 10/50/90/99/max:", np.percentile(cn[cn > 0], [10, 50, 90, 99, 100]).astype(int))
print("cluster size pct 10/50/90/99/max:", np.percentile(sizes_all, [10, 50, 90, 99, 100]).astype(int))
for lo, hi in ((1, 1), (2, 8), (9, 64), (65, 256), (257, 1024), (1025, 4096), (4097, 1 << 30)):
    k = (sizes_all >= lo) & (sizes_all <= hi)
    print("  clusters with %5d..%-10d members: %6d  (points %8d)" % (lo, hi, k.sum(), sizes_all[k].sum()))
order = torch.argsort(cnt, descending=True).tolist()
for s_ in order[:10]:
    one = torch.zeros_like(cnt); one[s_] = cnt[s_]
    t = timeit(lambda: fit(one))
    b0 = int(eng.base_a[s_]); n_ = int(cnt[s_])
    lab = eng.label[b0:b0 + n_]
    roots, sizes = torch.unique(lab, return_counts=True)
    print("seg %4d n=%6d clusters=%3d top %s  %.3f ms" % (s_, n_, roots.numel(), sorted(sizes.tolist())[::-1][:4], t))
small = cnt.clone(); small[cnt > 2048] = 0
print("only n<=2048: %.3f ms" % timeit(lambda: fit(small)))
none = torch.zeros_like(cnt)
print("empty: %.3f ms" % timeit(lambda: fit(none)))
