"""Dev tool (GPU): randomised segment layouts through dfu3d_radius_filter (FLAGS | RESOLVE, no compaction) against a
brute-force float64 count; prints the context of every flag that differs (position inside its 512-slot range, segment
extent, nearest neighbours).

    python tools/rf_fuzz.py [rounds=40] [seed=0]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from dfu3d_amd import stages as st

dev = "cuda:0"
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(dev)
bad_total = 0
for rnd in range(rounds):
    S = int(rng.integers(1, 40))
    segs, rad = [], []
    for s in range(S):
        kind = rng.integers(0, 5)
        n = int([rng.integers(0, 8), rng.integers(1, 70), rng.integers(30, 300), rng.integers(200, 1500), rng.integers(1000, 6000)][kind])
        scale = float(rng.choice([0.5, 2.0, 8.0, 30.0]))
        pts = rng.normal(0, scale, (n, 3))
        if n and rng.random() < 0.5:                       # surface-like runs: consecutive points close together
            pts = np.cumsum(rng.normal(0, 0.3, (n, 3)), 0)
            jump = rng.random(n) < 0.05
            pts[jump] += rng.uniform(-50, 50, (int(jump.sum()), 3))
        segs.append(pts)
        rad.append(float(rng.choice([3.0, 0.6, 0.6, 3.0, 0.0, -1.0])))
    nb = int(rng.choice([1, 1, 1, 2, 4]))
    base, cur = [], int(rng.integers(0, 600))
    chunks = [np.full((cur, 3), 12345.0)]
    for p_ in segs:
        base.append(cur)
        gap = int(rng.choice([0, 0, 0, 3, 64]))
        chunks += [p_, np.full((gap, 3), 777.0)]
        cur += len(p_) + gap
    P = np.concatenate(chunks)
    cap = len(P) + 1024
    Pp = np.full((cap, 3), 555.0); Pp[:len(P)] = P
    px, py, pz = t(Pp[:, 0]), t(Pp[:, 1]), t(Pp[:, 2])
    cnt = np.array([len(p_) for p_ in segs], np.int32)
    flags = torch.full((cap,), 7, dtype=torch.uint8, device=dev)
    st.radius_filter(px, py, pz, t(np.array(base, np.int64)), t(cnt), t(np.array(rad)), nb, S, cap,
                     torch.zeros(S + 1, dtype=torch.int32, device=dev), flags,
                     torch.zeros(st.rf_queue_ints(cap), dtype=torch.int32, device=dev), phases=st.RF_ALL & ~st.RF_COMPACT)
    torch.cuda.synchronize()
    F = flags.cpu().numpy()
    for s, pts in enumerate(segs):
        n = len(pts)
        if n == 0:
            continue
        r = rad[s]
        if r == 0.0:
            exp = np.ones(n, bool)
        elif r < 0:
            exp = np.zeros(n, bool)
        else:
            d2 = ((pts[:, None, :] - pts[None, :, :]) ** 2)
            d2 = (d2[:, :, 0] + d2[:, :, 1]) + d2[:, :, 2]
            exp = (d2 < r * r).sum(1) > nb
        got = F[base[s]:base[s] + n]
        for i in np.nonzero((got != 0) != exp)[0]:
            bad_total += 1
            pos = base[s] + i
            near = np.sort(np.sqrt(d2[i]))[:6] if r > 0 else []
            who = np.argsort(d2[i])[:6] if r > 0 else []
            print("round %d nb %d seg %d (n %d, base %d = range %d slot %d, r %.1f): point %d at pool %d (range %d, chunk %d, lane %d) "
                  "flag %d expected %d; nearest d %s at list offsets %s"
                  % (rnd, nb, s, n, base[s], base[s] // 512, base[s] % 512, r, i, pos, pos // 512, (pos % 512) // 64, pos % 64,
                     got[i], int(exp[i]), np.round(near, 4), [int(w - i) for w in who]), flush=True)
print("rf_fuzz: %d rounds, %d wrong flags" % (rounds, bad_total))
