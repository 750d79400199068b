"""Dev tool (GPU): A/B of tuning builds of the library in ONE process on ONE box -- box-to-box and run-to-run differences of a
stage (k_bp_vox: 1.07 .. 1.33 ms between boxes) are larger than most tuning effects, so two builds are only comparable when
they alternate inside one run.

    python tools/ab_builds.py [--frames 64] [--rounds 3] [--passes 4] name=-DFLAG[,-DFLAG2] ...

Every `name=flags` is compiled from the tracked sources with the extra flags into csrc/libdfu3d_hip_ab_<name>.so (git-ignored,
travels to the GPU box with the snapshot; delete them afterwards); `product` stands for the library as shipped.  For every round and build: one engine,
one untimed pass, `passes` passes with every stage bracketed by HIP events (single stream, 384 views per launch); prints
ms per pass and stage, the mean over the rounds, and whether the rows of each build equal the first build's.
Compile on the build container first (`--compile-only`): the GPU box then finds the .so files in the snapshot."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--passes", type=int, default=4)
    ap.add_argument("--compile-only", action="store_true")
    ap.add_argument("--throughput", action="store_true",
                    help="instead of the stage table: frames/s of the timed region's form (2 free lanes x 32-frame chunks, one C call per "
                         "chunk, pass i+1 enqueued before pass i is collected under a side stream), wall clock over --passes passes")
    ap.add_argument("builds", nargs="+")
    args = ap.parse_args()
    from dfu3d_amd import _build
    libs = []
    for spec in args.builds:
        name, _, flags = spec.partition("=")
        if name == "product":
            libs.append((name, _build.build()))
            continue
        out = os.path.join(_build.CSRC, "libdfu3d_hip_ab_%s.so" % name)
        if not os.path.exists(out) or _build._stale(out):
            _build._compile(out, {"force": True, "flags": [f for f in flags.split(",") if f]}, False)
        libs.append((name, out))
    if args.compile_only:
        for n, p in libs:
            print(n, p)
        return
    import ctypes
    import torch
    from dfu3d_amd import _lib, synth
    from dfu3d_amd.engine import PseudoBoxEngine
    from dfu3d_amd.params import Params
    dev = "cuda:0"
    p = Params()
    scenes = [synth.make_scene(f, dense=True, device=dev, k_min=30, k_max=40) for f in range(args.frames)]
    b = synth.to_view_batch(scenes, p, dev)
    b.pack_masks()                                   # (as `python bench.py`: the masks resident as bit-packed words)
    loaded = {}
    for name, path in libs:
        L = ctypes.CDLL(path)
        for sym, (res, a) in _lib.SIGNATURES.items():
            fn = getattr(L, sym)
            fn.restype, fn.argtypes = res, a
        loaded[name] = L
    if args.throughput:
        import time
        res = {n: [] for n, _ in libs}
        ref = None
        for rnd in range(args.rounds):
            for name, _ in libs:
                _lib._LIB = loaded[name]
                eng = PseudoBoxEngine(p, 900, 1600, 8, 34720, views_per_chunk=args.frames * 6 // 2, dense=True, cap_vox=1 << 18,
                                      pool_per_view=1 << 17, device=dev, lanes=2, chain=True, free_lanes=True)
                side = torch.cuda.Stream()

                def loop(k):
                    pending, out = None, None
                    for _ in range(k):
                        h = eng.launch(b)
                        if pending is not None:
                            with torch.cuda.stream(side):
                                out = eng.collect(pending)
                            side.synchronize()
                        pending = h
                    with torch.cuda.stream(side):
                        out = eng.collect(pending)
                    side.synchronize()
                    return out
                rows, st_ = loop(3)
                assert int(st_) == 0
                if ref is None:
                    ref = rows.clone()
                same = rows.shape == ref.shape and bool(torch.equal(rows, ref))
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                loop(args.passes)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                fps = args.frames * args.passes / dt
                res[name].append(fps)
                print("round %d %-12s %s %.0f frames/s (%.3f ms per %d frames)" % (rnd, name, "rows_equal" if same else "ROWS DIFFER", fps,
                                                                                   dt / args.passes * 1e3, args.frames), flush=True)
                del eng
                torch.cuda.empty_cache()
        for name, _ in libs:
            v = res[name]
            print("  %-12s mean %.0f  min %.0f  max %.0f frames/s" % (name, sum(v) / len(v), min(v), max(v)))
        return
    ref_rows = None
    differ = set()
    acc = {n: {} for n, _ in libs}
    for rnd in range(args.rounds):
        for name, _ in libs:
            _lib._LIB = loaded[name]
            eng = PseudoBoxEngine(p, 900, 1600, 8, 34720, views_per_chunk=args.frames * 6, dense=True, cap_vox=1 << 18,
                                  pool_per_view=1 << 17, device=dev, lanes=1, chain=False)
            rows, st_ = eng.run(b)
            assert int(st_) == 0, int(st_)
            rows = rows[torch.argsort(rows[:, 0] * 1e6 + rows[:, 1] * 1e3 + rows[:, 2])] if rows.numel() else rows
            if ref_rows is None:
                ref_rows = rows.clone()
            same = rows.shape == ref_rows.shape and bool(torch.equal(rows, ref_rows))
            if not same:
                differ.add(name)
            eng.reset_timing()
            eng.timing = True
            for _ in range(args.passes):
                eng.run(b)
            eng.timing = False
            t = eng.timing_summary()
            per = {k: v[0] / args.passes for k, v in t.items()}
            tot = sum(per.values())
            for k, v in per.items():
                acc[name].setdefault(k, []).append(v)
            acc[name].setdefault("pass", []).append(tot)
            print("round %d %-12s %s pass %.3f  %s" % (rnd, name, "rows_equal" if same else "ROWS DIFFER", tot,
                                                      " ".join("%s %.4f" % (k, v) for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:9])),
                  flush=True)
            del eng
            torch.cuda.empty_cache()
    print("mean over the rounds after the first:")
    keys = sorted(acc[libs[0][0]], key=lambda k: -sum(acc[libs[0][0]][k]))
    for name, _ in libs:
        m = {k: sum(v[1:]) / max(len(v) - 1, 1) for k, v in acc[name].items()}
        print("  %-12s pass %.3f  %s" % (name, m["pass"], " ".join("%s %.4f" % (k, m[k]) for k in keys if k != "pass")[:400]))
    if differ:                                        # (say it LAST: a tuning build with other rows is a bug, whatever its speed)
        print("ROWS DIFFER from the first build's in: " + ", ".join(sorted(differ)))
        sys.exit(1)
    print("rows of every build equal")


if __name__ == "__main__":
    main()
