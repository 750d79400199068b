"""Dev tool (GPU): the radius-filter stage on the state the engine hands it (joint LiDAR + pseudo lists of `frames` bench
frames): time per kernel (HIP events on the launch stream, best / median of `reps`), queue length after phase A, length of
the U lists, achieved GB/s of the stage at 21 B/point, and a check of the keep mask against a brute-force float64 count on
a sample of segments (every segment of one frame: tests/test_gpu_radius_filter.py).

    python tools/profile_rf.py [frames=64] [reps=20]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from dfu3d_amd import stages as st, synth
from dfu3d_amd.engine import PseudoBoxEngine
from dfu3d_amd.params import Params

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = "cuda:0"
p = Params()
scenes = [synth.make_scene(f, dense=True, device=dev, k_min=30, k_max=40) for f in range(frames)]
b = synth.to_view_batch(scenes, p, dev)
eng = PseudoBoxEngine(p, 900, 1600, 8, 34720, views_per_chunk=frames * 6)
snaps = []
orig = st.radius_filter


def hook(px, py, pz, base, cnt, radius, nb, S, pool_cap, tile_off, flags, queue, phases=st.RF_ALL, shadow=None, n_used=None):
    snaps.append(dict(px=px.clone(), py=py.clone(), pz=pz.clone(), base=base.clone(), cnt=cnt.clone(), radius=radius.clone(),
                      S=S, phases=phases, shadow=shadow.clone(), n_used=n_used.clone()))
    return orig(px, py, pz, base, cnt, radius, nb, S, pool_cap, tile_off, flags, queue, phases=phases, shadow=shadow, n_used=n_used)


st.radius_filter = hook
eng.run(b)
torch.cuda.synchronize()
st.radius_filter = orig
joint = snaps[0]                    # FLAGS | RESOLVE over the 2S joint lists
n_pts = int(joint["cnt"].sum())
S2 = joint["S"]


def timed(fn):
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return min(ts), float(np.median(ts))


def call(sn, phases, cnt=None):
    orig(sn["px"], sn["py"], sn["pz"], sn["base"], sn["cnt"] if cnt is None else cnt, sn["radius"], p.nb_points, sn["S"],
         eng.pool_cap, eng.tile_off, eng.flags, eng.queue, phases=phases, shadow=sn["shadow"], n_used=sn["n_used"])


t_flags = timed(lambda: call(joint, st.RF_FLAGS))
nq = int(eng.queue[0:1024:16].sum())
t_res = timed(lambda: call(joint, st.RF_RESOLVE))
flags = eng.flags.clone()
stage = t_flags[0] + t_res[0]
print("points %d  segments %d  queued %d (%.3f%%)" % (n_pts, S2, nq, 100.0 * nq / max(n_pts, 1)))
print("phase A %.1f / %.1f us   phase B %.1f / %.1f us   (best / median, events include ~2 us launch; the lists are compacted by the fuse)"
      % (t_flags + t_res))
print("stage (best) %.1f us = %.0f GB/s at 21 B/point = %.3f of 8 TB/s; phase A alone %.0f GB/s" %
      (stage, 21.0 * n_pts / stage / 1e3, 21.0 * n_pts / stage / 1e3 / 8000.0, 21.0 * n_pts / t_flags[0] / 1e3))
# U slots: the count sits next to each range's first box
n_ranges = (eng.pool_cap + 511) // 512 + 1
sh = joint["shadow"]
rec = sh[4 * eng.pool_cap:4 * eng.pool_cap + 16 * n_ranges].view(-1, 16)
used = (int(joint["n_used"]) + 511) // 512
nl = rec[:used, 3].contiguous().view(torch.int32).cpu().numpy()
uoff = 4 * eng.pool_cap + (16 + 512 + 128) * n_ranges + 4 * eng.pool_cap + 32
ucount = sh[uoff:uoff + S2].view(torch.int32).cpu().numpy()
print("listed points: %d (%.2f%% of the pool), fullest range %d of 128; overflow lists %d points" %
      (nl.sum(), 100.0 * nl.sum() / max(n_pts, 1), nl.max(), ucount.sum()))
# keep mask against a brute-force float64 count (sample of segments, all sizes)
base = joint["base"].cpu().numpy(); cnt = joint["cnt"].cpu().numpy(); rad = joint["radius"].cpu().numpy()
X = torch.stack([joint["px"], joint["py"], joint["pz"]], 1)
rng = np.random.default_rng(0)
order = [s for s in np.argsort(-cnt)[:3]] + list(rng.permutation(np.nonzero(cnt > 0)[0])[:200])
bad = tot = 0
for s in order:
    n = int(cnt[s]);
    if n == 0:
        continue
    P = X[base[s]:base[s] + n]
    r = float(rad[s])
    if r == 0.0:
        exp = torch.ones(n, dtype=torch.bool, device=dev)
    elif not r > 0.0:
        exp = torch.zeros(n, dtype=torch.bool, device=dev)
    else:
        exp = torch.empty(n, dtype=torch.bool, device=dev)
        for q0 in range(0, n, 4096):
            Q = P[q0:q0 + 4096]
            c = torch.zeros(Q.shape[0], dtype=torch.int64, device=dev)
            for j0 in range(0, n, 16384):
                R = P[j0:j0 + 16384]
                dx = Q[:, None, 0] - R[None, :, 0]; dy = Q[:, None, 1] - R[None, :, 1]; dz = Q[:, None, 2] - R[None, :, 2]
                d = dx * dx; d = d + dy * dy; d = d + dz * dz
                c += (d < r * r).sum(1)
            exp[q0:q0 + 4096] = c > p.nb_points
    got = flags[base[s]:base[s] + n] != 0
    bad += int((got != exp).sum()); tot += n
print("keep mask vs brute force: %d differences in %d points of %d segments" % (bad, tot, len(order)))
