"""Randomised engine-vs-oracle parity sweep (GPU box): many seeds, image sizes, chunkings, lanes, mask layouts, call
paths and parameter variants (nb_points, voxel caps, the statistical pair on, sparse mode, H4 off).  A fixed-seed slice
of it is in the pytest suite (tests/test_gpu_engine.py::test_seeded_slice_of_the_randomised_parity_sweep); run more
after kernel changes:  python tools/stress_parity.py [n_cases] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def run_cases(n_cases=24, seed=2026, budget_s=None, verbose=False):
    """-> (cases run, mismatches, log lines).  budget_s: stop starting new cases after that many seconds."""
    from dfu3d_amd import synth
    from dfu3d_amd.engine import PseudoBoxEngine
    from dfu3d_amd.params import Params
    from tests.test_gpu_engine import _oracle_rows, _compare
    rng = np.random.default_rng(seed)
    bad, done, log = 0, 0, []
    t0 = time.time()
    for case in range(n_cases):
        if budget_s is not None and time.time() - t0 > budget_s:
            break
        H, W = [(180, 320), (225, 400), (120, 480), (300, 400)][case % 4]
        cams = int(rng.integers(1, 5))
        frames = int(rng.integers(1, 4))
        M = int(rng.choice([3, 6, 8]))
        dense = bool(case % 5 != 4)
        kw = {}
        variant = case % 6
        if variant == 1: kw["nb_points"] = 2
        if variant == 2: kw["max_points_per_voxel"] = 3
        if variant == 3: kw["stat_filter"] = True
        if variant == 4: kw["small_class_drop_lidar"] = False
        if variant == 5: kw["max_voxels"] = 4000
        p = Params(bounds_hw=(H, W), fov_hw=(H, W), **kw)
        sseed = int(rng.integers(1000, 100000))
        scenes = [synth.make_scene(sseed + f, H=H, W=W, M=M, cams=cams, dense=dense, k_min=8, k_max=22,
                                   rings=int(rng.choice([16, 32])), az=int(rng.choice([400, 1085]))) for f in range(frames)]
        b = synth.to_view_batch(scenes, p, "cuda:0", dense=dense)
        packed = bool(rng.integers(0, 2))
        if packed:
            b.pack_masks()
        cap_n = max(s.points.shape[0] for s in scenes)
        V = cams * frames
        vpc = int(rng.choice([d for d in range(1, V + 1) if V % d == 0]))
        lanes = int(rng.integers(1, 4))
        chain = bool(rng.integers(0, 2))
        eng = PseudoBoxEngine(p, H, W, M, cap_n, views_per_chunk=vpc, dense=dense, cap_vox=1 << 18, pool_per_view=1 << 18,
                              lanes=lanes, chain=chain)
        rows, status = eng.run(b)
        exp, _ = _oracle_rows(scenes, p, dense)
        what = ("%dx%d cams %d frames %d M %d dense %d %s vpc %d lanes %d chain %d packed %d"
                % (H, W, cams, frames, M, dense, kw, vpc, lanes, chain, packed))
        try:
            assert status == 0, status
            _compare(rows, exp)
            rows2, _ = eng.run(b)
            assert torch.equal(rows, rows2)
            log.append("case %2d ok: %s -> %d boxes (%.0fs)" % (case, what, len(exp), time.time() - t0))
        except AssertionError as e:
            bad += 1
            log.append("case %2d MISMATCH seed %d %s: %s" % (case, sseed, what, str(e)[:300]))
        if verbose:
            print(log[-1], flush=True)
        done += 1
        del eng
    return done, bad, log


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    sd = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
    done, bad, _ = run_cases(n, sd, verbose=True)
    print("done: %d cases, %d mismatches" % (done, bad))
    sys.exit(1 if bad else 0)
