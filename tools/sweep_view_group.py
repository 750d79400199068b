"""Dev tool (GPU box): the back-projection's view-group size (dfu3d_backproject_bin: view_group) on the bench workload.

    python tools/sweep_view_group.py [frames=64] [G ...]

Per G: (a) one stream, one chunk of all frames, every kernel of every group bracketed with HIP events (the engine's
timing mode): back-projection kernels summed per pass; (b) the throughput configuration (2 streams x half the frames,
one C call per chunk): frames/s over 10 steps.  One JSON line per G on stdout."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from dfu3d_amd import synth  # noqa: E402
from dfu3d_amd.engine import PseudoBoxEngine  # noqa: E402
from dfu3d_amd.params import Params  # noqa: E402

dev = torch.device("cuda", 0)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 64
groups = [int(a) for a in sys.argv[2:]] or [0, 4, 8, 12, 16, 24, 32, 48]
params = Params()
H, W, M, CAMS, N_PTS = 900, 1600, 8, 6, 34720
scenes = [synth.make_scene(f, H=H, W=W, M=M, cams=CAMS, dense=True, device=dev, k_min=30, k_max=40) for f in range(frames)]
batch = synth.to_view_batch(scenes, params, dev, dense=True, frame_ids=list(range(frames)))
batch.pack_masks()
del scenes
torch.cuda.synchronize()


def engine(lanes, vpc, chain):
    return PseudoBoxEngine(params, H, W, M, N_PTS, views_per_chunk=vpc, dense=True, cap_vox=1 << 18,
                           pool_per_view=1 << 17, device=dev, lanes=lanes, chain=chain)


def set_group(eng, G):
    eng.view_group = G
    for L in eng.lanes:
        if eng.chain:
            L.chain_cfg.bp_view_group = G


ref = None
res = {}
# (a) per-kernel, one stream
eng = engine(1, frames * CAMS, False)
for G in groups:
    set_group(eng, G)
    rows, st_ = eng.run(batch)
    assert st_ == 0, st_
    if ref is None:
        ref = rows.clone()
    else:
        assert torch.equal(rows, ref), "rows differ at G=%d" % G
    eng.reset_timing()
    eng.timing = True
    for _ in range(3):
        eng.run(batch)
    eng.timing = False
    k = eng.timing_summary()
    res[G] = {"G": G, "ms_per_pass": {n: round(ms / 3, 4) for n, (ms, _) in sorted(k.items()) if n.startswith("bp_")},
              "pass_ms": round(sum(ms for ms, _ in k.values()) / 3, 3)}
del eng
torch.cuda.empty_cache()
# (b) throughput configuration
eng = engine(2, frames * CAMS // 2, True)
for G in groups:
    set_group(eng, G)
    for _ in range(3):
        eng.run(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pending = None
    for _ in range(10):
        h = eng.launch(batch)
        if pending is not None:
            eng.collect(pending)
        pending = h
    r_, st_ = eng.collect(pending)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert st_ == 0 and torch.equal(r_, ref)
    res[G]["frames_per_s_2x%d" % (frames // 2)] = round(10 * frames / dt, 1)
    print(json.dumps(res[G]), flush=True)
