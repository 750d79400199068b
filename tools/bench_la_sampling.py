"""Measurement of SURVEY.md §8 row f-4 on the GPU box: la_sampling of a frame's sampled objects (one launch) next to the
NumPy restatement of the reference's per-point Python loop.  One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dfu3d_amd.pcdet_kitti.database_sampler_virtual import la_sampling_batch
from oracle import la_sampling_oracle as LA

rng = np.random.default_rng(3)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 40
objs = []
for k in range(B):
    n = int(rng.integers(200, 3000))
    d, a = rng.uniform(5, 50), rng.uniform(-3.1, 3.1)
    p = np.zeros((n, 8), np.float32)
    p[:, :3] = (np.array([d * np.cos(a), d * np.sin(a), -0.8]) + rng.normal(0, [0.9, 0.45, 0.4], (n, 3))).astype(np.float32)
    objs.append(p)
tobjs = [torch.from_numpy(o).cuda() for o in objs]
for _ in range(3):
    got = la_sampling_batch(tobjs)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    got = la_sampling_batch(tobjs)
torch.cuda.synchronize()
gpu_ms = (time.perf_counter() - t0) / 10 * 1e3
t0 = time.perf_counter()
exp = [LA.la_sampling(o) for o in objs[:8]]
cpu_ms = (time.perf_counter() - t0) * 1e3 * B / 8
same = sum(int(g.shape == e.shape and np.array_equal(g.cpu().numpy(), e)) for g, e in zip(got[:8], exp))
pts = sum(len(o) for o in objs)
print(json.dumps({"objects": B, "points": pts, "gpu_ms_per_batch": round(gpu_ms, 3), "points_per_s": round(pts / (gpu_ms * 1e-3), 0),
                  "kept_rows": int(sum(len(g) for g in got)), "identical_to_oracle_of_first_8": same,
                  "cpu_baseline": {"ms_per_batch": round(cpu_ms, 1), "cores": 1, "kind": "port",
                                   "sample": "8 of the %d objects through oracle/la_sampling_oracle (the reference's per-point loop), scaled" % B}}))
