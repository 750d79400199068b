"""Dev tool (GPU): phases of dfu3d_radius_filter on the state snapshotted before each of the engine's two calls
(LiDAR lists, pseudo lists): time per phase, queue length after phase A, achieved GB/s of 21 B/point."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dfu3d_amd import synth, stages as st
from dfu3d_amd.engine import PseudoBoxEngine
from dfu3d_amd.params import Params
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = "cuda:0"; p = Params()
scenes = [synth.make_scene(f, dense=True, device=dev, k_min=30, k_max=40) for f in range(frames)]
b = synth.to_view_batch(scenes, p, dev)
eng = PseudoBoxEngine(p, 900, 1600, 8, 34720, views_per_chunk=frames * 6, pool_per_view=1 << 17)
snaps = []
orig = st.radius_filter
def hook(px, py, pz, base, cnt, radius, nb, S, pool_cap, tile_off, flags, queue, phases=st.RF_ALL):
    snaps.append(dict(px=px.clone(), py=py.clone(), pz=pz.clone(), base=base.clone(), cnt=cnt.clone(), radius=radius.clone()))
    return orig(px, py, pz, base, cnt, radius, nb, S, pool_cap, tile_off, flags, queue, phases=phases)
st.radius_filter = hook
eng.run(b); torch.cuda.synchronize()
st.radius_filter = orig
S = eng.Vc * eng.M
for name, sn in zip(("lidar", "pseudo"), snaps):
    n = int(sn["cnt"].sum())
    best = {}
    for rep in range(4):
        px, py, pz = sn["px"].clone(), sn["py"].clone(), sn["pz"].clone()
        base, cnt = sn["base"].clone(), sn["cnt"].clone()
        for tag, ph in (("tiles", st.RF_TILES), ("flags", st.RF_FLAGS), ("resolve", st.RF_RESOLVE), ("compact", st.RF_COMPACT)):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            orig(px, py, pz, base, cnt, sn["radius"], p.nb_points, S, eng.pool_cap, eng.tile_off, eng.flags, eng.queue, phases=ph)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
            best[tag] = min(best.get(tag, 1e9), dt)
            if tag == "flags" and rep == 0:
                nq = int(eng.queue[0])
            if tag == "resolve" and rep == 0:
                nlong = int(eng.queue[1])
    kept = int(cnt.sum())
    print("%-6s points %8d kept %8d queued %7d (%.2f%%) long %6d  tiles %.3f flags %.3f resolve %.3f compact %.3f ms (host-timed, +~0.02 launch)  flags: %.0f GB/s"
          % (name, n, kept, nq, 100.0 * nq / max(n, 1), nlong, best["tiles"], best["flags"], best["resolve"], best["compact"],
             21.0 * n / (best["flags"] - 0.0) / 1e6))
    cn = sn["cnt"].cpu().numpy()
    print("       segments nonempty %d, size pct 50/90/99/max %s" % ((cn > 0).sum(), np.percentile(cn[cn > 0], [50, 90, 99, 100]).astype(int)))
