"""Dev tool (GPU): per-phase cycle counts of cluster_body for one big segment
(needs a library built with -DDFU3D_DBG_CLUSTER_TIMING)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dfu3d_amd import stages as st
d = np.load(sys.argv[1] if len(sys.argv) > 1 else "tools/slow_segments.npz")
for key in ("seg18", "seg123"):
    xy = d[key]; n = len(xy); cap = n + 64; dev = "cuda:0"
    px = torch.zeros(cap, dtype=torch.float64, device=dev); py = px.clone()
    px[:n] = torch.from_numpy(xy[:, 0]); py[:n] = torch.from_numpy(xy[:, 1])
    base = torch.zeros(1, dtype=torch.int64, device=dev); cnt = torch.full((1,), n, dtype=torch.int32, device=dev)
    label = torch.zeros(cap, dtype=torch.int32, device=dev)
    si = torch.zeros(3 * cap, dtype=torch.int32, device=dev)
    sx = torch.zeros(cap, dtype=torch.float64, device=dev); sy = sx.clone()
    for _ in range(2):
        st.range_cluster(px, py, base, cnt, 1, 3.0, 0.001, label, cap, sx, sy, si)
    torch.cuda.synchronize()
    raw = si[cap:cap + 2 * 8 * 8 * 2].cpu().numpy().view(np.int64)
    t = raw[:64].reshape(8, 8); c = raw[64:128].reshape(8, 8)
    names = ["init+boxes", "flatten", "summaries", "phaseA", "barrier", "phaseB", "final-flatten", "epilogue"]
    print(key, "n =", n)
    for w in (0, 3, 7):
        tot = t[w].sum()
        print("  wave %d: total %.2f Mcyc  " % (w, tot / 1e6) + "  ".join("%s %.0fk" % (nm, v / 1e3) for nm, v in zip(names, t[w])))
        print("          blocks %d descended %d boxtests(lane0) %d slowgroups %d slowiters %d unites(lane0) %d earlyexit %d" % tuple(c[w][:7]))
