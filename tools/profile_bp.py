"""Dev tool (GPU): phases of dfu3d_backproject_bin on a synthetic batch + ambiguous-pixel fraction."""
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
V = eng.Vc; H, W = 900, 1600
status = torch.zeros(1, dtype=torch.int32, device=dev)
def run(ph):
    st.backproject_bin(b.depth, b.calib, b.masks, b.n_inst, V, eng.M, H, W, eng.geom, eng.E, 1, eng.table, eng.pix_bin,
                       eng.blk_cnt, eng.cap_vox, eng.n_vox, eng.vox_pix, eng.b_bits, eng.b_x, eng.b_y, eng.b_z, status, phases=ph)
def t_all():
    for ph in (1, 2, 8, 16): run(ph)
t_all(); torch.cuda.synchronize()
nblk = (H * W + 1023) // 1024
# n_amb is reset only by the next BIN phase -> still readable
amb = eng.blk_cnt[2 * V * (nblk + 1) + 2 * V: 2 * V * (nblk + 1) + 3 * V].cpu().numpy()
kept = (eng.pix_bin[:V * H * W] != -1).view(V, -1).sum(1).cpu().numpy()
print("ambiguous per view:", amb[:6], "kept per view:", kept[:6], "frac amb of kept: %.4f" % (amb.sum() / kept.sum()))
for name, ph in (("bin", 1), ("repair", 2), ("emit", 8), ("vox", 16)):
    ts = []
    for _ in range(3):
        for q in (1, 2, 8, 16):
            if q == ph:
                torch.cuda.synchronize(); t0 = time.perf_counter(); run(q); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            else:
                run(q)
    print("%-7s %.3f ms  (%.1f GB/s of depth)" % (name, min(ts) * 1e3, V * H * W * 4 / min(ts) / 1e9))
