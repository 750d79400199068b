"""Dev tool (GPU box): time phase A of the radius filter (k_rf_stream) in several builds of the library, each in a
process of its own (DFU3D_LIB_VARIANT is read at load time): HIP events around 20 back-to-back launches on the bench state.

    python tools/rf_variants.py [variant ...]        (no argument: the product build)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    from dfu3d_amd import stages as st, synth
    from dfu3d_amd.engine import PseudoBoxEngine
    from dfu3d_amd.params import Params
    frames = 64
    dev = "cuda:0"
    p = Params()
    scenes = [synth.make_scene(f, dense=True, device=dev, k_min=30, k_max=40) for f in range(frames)]
    b = synth.to_view_batch(scenes, p, dev)
    b.pack_masks()
    eng = PseudoBoxEngine(p, 900, 1600, 8, 34720, views_per_chunk=frames * 6)
    snaps = []
    orig = st.radius_filter

    def hook(*a, **k):
        snaps.append((a, k))
        return orig(*a, **k)
    st.radius_filter = hook
    eng.run(b)
    torch.cuda.synchronize()
    st.radius_filter = orig
    a, k = snaps[0]
    n_pts = int(a[4].sum())
    for phases, name in ((st.RF_FLAGS, "phase A"), (st.RF_RESOLVE, "phase B")):
        k2 = dict(k); k2["phases"] = phases
        for _ in range(3):
            orig(*a, **k2)
        torch.cuda.synchronize()
        reps = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            orig(*a, **k2)
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print("%-20s %-8s %7.1f us per call (back to back, memsets included)  %6.0f GB/s at 21 B/point" %
              (os.environ.get("DFU3D_LIB_VARIANT", "product"), name, us, 21.0 * n_pts / us / 1e3), flush=True)
    # the short-list compaction (third call of the engine: the LiDAR lists), counts restored before every launch
    a3, k3 = snaps[1] if len(snaps) > 1 else (None, None)
    if a3 is not None:
        cnt0 = a3[4].clone()
        reps = 20
        ts = []
        for _ in range(reps + 3):
            a3[4].copy_(cnt0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            orig(*a3, **k3)
            e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        print("%-20s %-8s %7.1f us per call (best of %d, one launch between events)" %
              (os.environ.get("DFU3D_LIB_VARIANT", "product"), "compact", min(ts[3:]), reps), flush=True)
    sys.exit(0)
for v in (sys.argv[1:] or ["product"]):
    env = dict(os.environ)
    if v != "product":
        env["DFU3D_LIB_VARIANT"] = v
    else:
        env.pop("DFU3D_LIB_VARIANT", None)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
    out = [l for l in r.stdout.splitlines() if "us per call" in l]
    print("\n".join(out) if out else "%s: failed\n%s" % (v, r.stderr[-800:]), flush=True)
