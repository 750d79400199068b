"""Row f-3, AP evaluation: time of get_official_eval_result on a synthetic split (dev tool, not the headline bench).

    python tools/bench_eval.py [frames] [classes]

Prints the wall time of the whole call (host packing, three metrics x two device passes, threshold picking on the
host), the device time of the kernels (HIP events around the launches of one metric), and the oracle's plain-Python
loops on a bounded sample of the same frames as the CPU figure (the reference runs these loops under numba, which this
image lacks -- the figure is a port's, not the reference's)."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests', 'golden'))
import eval_cases  # noqa: E402

from dfu3d_amd import stages as st  # noqa: E402
from dfu3d_amd.pcdet_kitti import eval as E  # noqa: E402


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    n_cls = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    classes = ['Car', 'Pedestrian', 'Truck', 'Bus', 'Bicycle', 'Barrier', 'Motorcycle', 'Trailer', 'Traffic_cone',
               'Construction_vehicle'][:n_cls]
    t0 = time.time()
    gts, dts = eval_cases.make_annos(1, frames, max_gt=30, extra_dt=40, classes=tuple(classes))
    n_gt, n_dt = sum(len(g['name']) for g in gts), sum(len(d['name']) for d in dts)
    print("synthetic split: %d frames, %d ground truths, %d detections (%.1f s to make)" % (frames, n_gt, n_dt, time.time() - t0))
    E.get_official_eval_result(gts[:8], dts[:8], classes)                     # warm-up: library load, first launches
    torch.cuda.synchronize()
    t0 = time.time()
    text, ret = E.get_official_eval_result(gts, dts, classes)
    torch.cuda.synchronize()
    wall = time.time() - t0
    print("get_official_eval_result: %.3f s  (%d classes x 3 levels x 3 overlaps x 3 metrics)" % (wall, n_cls))
    t0 = time.time()
    ev = E._Evaluator(gts, dts)
    torch.cuda.synchronize()
    print("  host packing + upload: %.3f s" % (time.time() - t0))
    cls = [E.CLASS_NAMES.index(c) for c in classes]
    mo = E.official_min_overlaps()[:, :, cls]
    torch.rand((4, 41), dtype=torch.float64, device='cuda').sort(dim=1, descending=True)        # (torch's own first-call set-up)
    torch.cuda.synchronize()
    for metric in (0, 1, 2):
        ev0, ev1, ev2, ev3 = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev0.record()
        ov = ev.overlaps(metric)
        ev1.record()
        combos = [(c, l, float(mo[k, metric, m])) for m, c in enumerate(cls) for l in range(3) for k in range(3)]
        matched, n_valid = st.eval_match_scores(ev.fr, ov, metric, combos)
        th = torch.rand((len(combos), 41), dtype=torch.float64, device='cuda').sort(dim=1, descending=True)[0]
        nth = torch.full((len(combos),), 41, dtype=torch.int32, device='cuda')
        ev2.record()
        st.eval_match_stats(ev.fr, ov, metric, combos, th, nth, metric == 0)
        ev3.record()
        torch.cuda.synchronize()
        print("  metric %d: overlaps %.3f ms (%d pairs), matched scores %.3f ms, statistics %.3f ms (%d cells x 41 thresholds)"
              % (metric, ev0.elapsed_time(ev1), ev.fr.n_pairs, ev1.elapsed_time(ev2), ev2.elapsed_time(ev3), len(combos)))
    if '--no-cpu' not in sys.argv:
        from oracle import kitti_eval_oracle as KO
        sample = min(frames, 24)
        t0 = time.time()
        KO.official_result(gts[:sample], dts[:sample], classes)
        cpu = time.time() - t0
        print("oracle (plain-Python port, 1 core) on %d frames: %.2f s -> %.1f frames/s; GPU path: %.0f frames/s"
              % (sample, cpu, sample / cpu, frames / wall))
    print({k: round(float(v), 3) for k, v in list(ret.items())[:4]})


if __name__ == '__main__':
    main()
