#!/usr/bin/env python3
"""bench.py -- pseudo-box frames/s on synthetic 34k-point, 6x1600x900 scenes.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL)

A step = one pass of the whole hot path (FOV filter, plane RANSAC, label
inheritance, back-projection + spherical voxel sampling, radius filters,
BallQuery fuse, clustering, L-shape fit, box rows) over one batch of
`--frames` synthetic frames per GPU that is resident in HBM, followed for N > 1
by the all-gather of the box rows.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from dfu3d_amd import dist as D            # noqa: E402
from dfu3d_amd import synth                # noqa: E402
from dfu3d_amd.engine import PseudoBoxEngine  # noqa: E402
from dfu3d_amd.params import Params        # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); the copy rate of THIS box is measured below
H, W, CAMS, MAX_INST, N_PTS = 900, 1600, 6, 8, 34720


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# engine stage (as bracketed with HIP events) -> the kernels rocprofv3 lists for it
STAGE_KERNELS = {
    "fov_filter": ["k_fov_filter"], "plane_ransac": ["k_plane_ransac"],
    "project_label": ["k_project_rows", "k_label_rows"],
    "bp_bin": ["k_bp_bin", "k_bp_prep", "k_bp_tables"], "bp_amb": ["k_bp_bin_amb"], "bp_mark": ["k_bp_scan"],
    "bp_vox": ["k_bp_vox"], "bp_repair": ["k_bp_rebin", "k_ovf_alloc", "k_ovf_gather", "k_ovf_select", "k_bp_fix",
                                          "k_bp_finalize"],
    "segments_build": ["k_seg_count", "k_seg_scan", "k_seg_alloc", "k_seg_write"],
    "rf_flags": ["k_rf_stream", "k_rf_pair"], "rf_resolve": ["k_rf_resolve", "k_rf_ranges"],
    "ballquery_fuse": ["k_tile_scan_class", "k_ball_flags", "k_seg_compact"],
    "range_cluster": ["k_range_cluster_grid", "k_range_cluster_small", "k_range_cluster_large"],
    "lshape_fit": ["k_fit_gather", "k_fit_tiny", "k_fit_medium", "k_fit_big_cost", "k_fit_big_box"],
}
RF_STAGE = ("rf_flags", "rf_resolve")     # (the compaction of both lists is the fuse's: one launch for both filters)


def algorithmic_bytes(c, views, max_inst, mask_word):
    """Compulsory HBM bytes of every stage for the bracketed passes (SURVEY.md 8d / DESIGN.md 5), from the
    device-side counts `c` of those passes: per unit figure x units."""
    px = views * H * W
    return {
        "fov_filter": 16 * views * N_PTS + 4 * c["fov_points"],                 # the sweep per view, index list out
        "plane_ransac": 20 * c["fov_points"] + 32 * views,                     # FOV points once, plane out
        "project_label": 20 * c["fov_points"] + (28 + (mask_word or max_inst)) * c["label_rows"],
        "bp_bin": 4 * px,                                                      # the depth map, once
        "bp_amb": 4 * c.get("amb_pixels", 0),
        "bp_mark": (4 * 32 * ((W + 63) // 64) * ((H + 15) // 16) + 4 * H * ((W + 63) // 64)) * views,   # bit map in, prefix out
        "bp_vox": (24 + 4 + (mask_word or max_inst) + 28) * c.get("voxels", 0),     # table entry, depth, mask, voxel record
        "bp_repair": 0,
        "segments_build": 4 * (c["label_rows"] + c.get("voxels", 0)) + (24 + 24) * c["pool_points"],
        "rf_stage": 21 * c["pool_points"],                                     # 16 coords + 4 segment + 1 flag
        "ballquery_fuse": 12 * c["ball_points"] + c["ball_points"],            # 8d S5: 12(p+l) + p (p+l counted)
        "range_cluster": 12 * c["instance_points"],                            # 8d S6: 8n + 4n
        "lshape_fit": 16 * c["instance_points"] + 72 * c["rows"],              # 8d S7
    }


def measure_copy_rate(dev, nbytes=1 << 30, reps=8):
    """float4 device-to-device copy on THIS GPU -> GB/s of (read + write) bytes (confirm the peak on the box)."""
    src = torch.empty(nbytes // 4, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    dst.copy_(src)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    del src, dst
    torch.cuda.empty_cache()
    return 2.0 * nbytes / (ms * 1e-3) / 1e9


def sources_sha16():
    """sha256 over the library's sources (csrc/*.hip, *.hpp, include/dfu3d.h), first 16 hex digits: the committed counter
    files carry the hash of the build they were collected on (tools/pmc_sq.sh, tools/pmc_to_json.py)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "dfu3d_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "dfu3d_amd", "csrc", "*.hpp"))
                    + [os.path.join(ROOT, "include", "dfu3d.h")]):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernels):
    """HBM bytes per launch of the given kernels from the newest committed rocprofv3 --pmc passes
    (profiles/rNN_pmc_traffic.json: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), summed; None if unknown.
    Valid for the launch geometry named in that file (bench default)."""
    import glob
    try:
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
        with open(files[-1]) as f:
            pm = json.load(f)
        tot = 0
        for k in kernels:
            hit = [v for name, v in pm["kernels"].items() if name == k or name.startswith(k + "<")]
            if not hit:
                continue
            for v in hit:
                if v["fetch_bytes"] is None or v["write_bytes"] is None:
                    return None, None
                tot += (v["fetch_bytes"] + v["write_bytes"]) * v.get("launches_per_pass", 1)
        return int(tot), {"file": os.path.basename(files[-1]), "views_per_launch": pm.get("views_per_launch"),
                          "collected_on_this_build": pm.get("sources_sha16") == sources_sha16()}
    except Exception:
        return None, None


def valu_limit(kernel, avg_ms):
    """What actually limits the dominant kernel, from the committed SQ counters (profiles/rNN_pmc_sq.json): the share
    of the launch during which the vector pipes issue (VALU wave-instructions x 4 cycles over 1024 SIMDs at 2.4 GHz).
    The roofline the contract prices a gather / scatter pass against is HBM (`bound`); a kernel far below it with
    busy vector pipes is issue-limited, and this field says so."""
    import glob
    try:
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_sq.json")))
        with open(files[-1]) as f:
            sq = json.load(f)
        v = sq.get(kernel)
        if not v:
            return None
        busy = v["SQ_INSTS_VALU"] * 4.0 / (1024 * 2.4e9) / (avg_ms * 1e-3)
        same = sq.get("_meta", {}).get("sources_sha16") == sources_sha16()
        return {"valu_busy_frac_est": round(busy, 3), "valu_insts_per_wave": round(v["valu_per_wave"], 1),
                "wait_any_frac": round(v["wait_any_frac"], 3), "source": os.path.basename(files[-1]),
                # the counters are from a committed profiling run, the duration from this one: only the same build makes a figure
                "collected_on_this_build": bool(same),
                "reading": (("vector-instruction issue, not HBM" if busy > 0.5 else "latency / memory") if same else
                            "counters of an OLDER build of the library: indicative only")}
    except Exception:
        return None


def cpu_baseline(scenes_cpu, params, dense, max_seconds=18.0):
    """The oracle (NumPy + C restatement, single thread) on a bounded sample of
    the same workload, on this box's host cores.  Its box rows are kept: they are what the `parity` block of the
    JSON line compares the rows of the TIMED engine with (frame f of this rank, camera c = view 6 f + c: the key of
    the seeded plane fit on both sides)."""
    from oracle import penet_oracle as O
    op = O.Params()
    t0 = time.time()
    done = 0
    exp = []
    for f, s in enumerate(scenes_cpu):
        pts = s.points.numpy()
        for c, cal in enumerate(s.calibs):
            oc = O.Calibration({"P2": cal.P2, "R0": cal.R0, "Tr_velo2cam": cal.V2C})
            lid, _ = O.fov_filter(pts, oc, params.fov_hw)
            n = int(s.n_inst[c])
            d = s.depth[c].numpy().copy() if dense else np.zeros((H, W), np.float32)
            res = O.depth2pointsrgbpm(d[:, :, None], None, oc, lid, O.NUSC_CLASSES,
                                      s.masks[c][:n].numpy().astype(np.float32),
                                      s.inst_class[c][:n].numpy(), s.inst_box[c][:n].numpy(), op,
                                      plane_key=f * CAMS + c, want_points=False)
            exp += [(f, c, r.inst, r.cluster, r.cls, r.as_vector()) for r in res.rows]
        done += 1
        if time.time() - t0 > max_seconds:
            break
    dt = time.time() - t0
    return done / dt, done, dt, exp


def parity_block(rows, exp, n_frames, frame_ids):
    """Rows of the timed engine (gathered layout: global frame, camera, class, instance, cluster, alpha, 2-D box,
    h w l, x y z, ry, score) against the oracle's rows for this rank's first n_frames frames: BASELINE.json's
    'box L1 vs ref' for the configuration that was timed (packed masks, chunked launches, streams, chain)."""
    R = rows.cpu().numpy()
    gid = {int(frame_ids[f]): f for f in range(n_frames)}
    sel = R[np.isin(R[:, 0].astype(np.int64), list(gid))]
    got = sorted(((gid[int(r[0])], int(r[1]), int(r[3]), int(r[4])), int(r[2]), r[5:17]) for r in sel)
    want = sorted(((f, c, j, k), cls, np.asarray(v, np.float64)) for f, c, j, k, cls, v in exp)
    same_keys = [g[0] for g in got] == [w[0] for w in want] and [g[1] for g in got] == [w[1] for w in want]
    out = {"frames": n_frames, "rows_oracle": len(want), "rows_gpu": len(got), "rows_equal": bool(same_keys),
           "how": "rows of the last timed step (this configuration: packed masks, chunks, streams) vs oracle/penet_oracle.py "
                  "on the same frames: (frame, camera, instance, cluster) keys and classes exact, 12 box numbers compared"}
    if same_keys and got:
        G = np.array([g[2] for g in got]); Wv = np.array([w[2] for w in want])
        d = np.abs(G - Wv)                       # alpha x1 y1 x2 y2 h w l x y z ry
        out.update(max_abs_centre_dims_m=float(d[:, 5:11].max()), max_abs_yaw_rad=float(d[:, 11].max()),
                   max_abs_alpha_rad=float(d[:, 0].max()), max_abs_bbox_px=float(d[:, 1:5].max()),
                   mean_abs_box_l1=float(d[:, 5:12].sum(1).mean()),
                   within_north_star_tolerance=bool(d[:, 5:11].max() <= 1e-3 and d[:, 11].max() <= 1e-3))
    return out


def _cpu_worker(go, barrier, tasks, done):
    """Forked before the GPU is touched; sleeps until the timed GPU region is over, then runs the oracle on one
    frame read from shared memory, together with its siblings (all-cores CPU figure)."""
    try:
        go.wait()
        t = tasks.get()
        if t is None:
            return
        from multiprocessing import shared_memory, resource_tracker
        import torch as _t
        _t.set_num_threads(1)
        from oracle import penet_oracle as O
        shm = shared_memory.SharedMemory(name=t["shm"])
        try:
            resource_tracker.unregister(shm._name, "shared_memory")
        except Exception:
            pass
        v = lambda key, shape, dt: np.ndarray(shape, dt, buffer=shm.buf, offset=t["off"][key])
        pts = v("points", t["n_pts"], np.float32).reshape(-1, 4)
        depth = v("depth", (CAMS, H, W), np.float32)
        masks = v("masks", (CAMS, MAX_INST, H, W), np.uint8)
        op = O.Params()
        barrier.wait(timeout=120)
        t0 = time.time()
        for c in range(CAMS):
            P2, R0, V2C = t["calib"][c]
            oc = O.Calibration({"P2": P2, "R0": R0, "Tr_velo2cam": V2C})
            lid, _ = O.fov_filter(pts, oc, t["fov_hw"])
            n = int(t["n_inst"][c])
            d = depth[c].copy() if t["dense"] else np.zeros((H, W), np.float32)
            O.depth2pointsrgbpm(d[:, :, None], None, oc, lid, O.NUSC_CLASSES, masks[c][:n].astype(np.float32),
                                t["classes"][c][:n], t["boxes"][c][:n], op, plane_key=c, want_points=False)
        done.put(time.time() - t0)
        shm.close()
    except Exception as e:                                   # never break the benchmark
        done.put("error: %s" % e)


def usable_cores():
    """Cores this process can actually use: its affinity mask, cut to the CPU quota of its control group (a GPU box
    gives a job a share of the host -- 16 cores per GPU on this pool -- while the mask still shows all of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    return n


def cpu_all_cores_start(nproc):
    """Fork the idle workers (must happen before the first GPU call of this process)."""
    import multiprocessing as mp
    ctx = mp.get_context("fork")
    st = {"go": ctx.Event(), "barrier": ctx.Barrier(nproc), "tasks": ctx.Queue(), "done": ctx.Queue(), "n": nproc}
    st["procs"] = [ctx.Process(target=_cpu_worker, args=(st["go"], st["barrier"], st["tasks"], st["done"]), daemon=True)
                   for _ in range(nproc)]
    for p_ in st["procs"]:
        p_.start()
    return st


def cpu_all_cores_run(st, scene_cpu, params, dense):
    """Every worker runs the oracle on the same frame at the same time -> frames/s of the whole host share."""
    from multiprocessing import shared_memory
    n = int(scene_cpu.n_inst.max())
    arrs = {"points": scene_cpu.points.numpy().astype(np.float32), "depth": scene_cpu.depth.numpy().astype(np.float32),
            "masks": scene_cpu.masks.numpy().astype(np.uint8)}
    off, cur = {}, 0
    for k, a in arrs.items():
        off[k] = cur
        cur += (a.nbytes + 4095) // 4096 * 4096
    shm = shared_memory.SharedMemory(create=True, size=cur)
    try:
        for k, a in arrs.items():
            np.ndarray(a.shape, a.dtype, buffer=shm.buf, offset=off[k])[...] = a
        task = {"shm": shm.name, "off": off, "n_pts": (arrs["points"].size,), "fov_hw": tuple(params.fov_hw), "dense": dense,
                "calib": [(c.P2, c.R0, c.V2C) for c in scene_cpu.calibs], "n_inst": scene_cpu.n_inst.numpy(),
                "classes": scene_cpu.inst_class.numpy(), "boxes": scene_cpu.inst_box.numpy()}
        for _ in range(st["n"]):
            st["tasks"].put(task)
        st["go"].set()
        res = [st["done"].get(timeout=180) for _ in range(st["n"])]
    finally:
        for p_ in st["procs"]:
            p_.join(timeout=5)
        shm.close()
        shm.unlink()
    if any(isinstance(r, str) for r in res):
        raise RuntimeError([r for r in res if isinstance(r, str)][0])
    return st["n"] / max(res), max(res)


def cpu_all_cores_cancel(st):
    for _ in range(st["n"]):
        st["tasks"].put(None)
    st["go"].set()
    for p_ in st["procs"]:
        p_.join(timeout=5)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes through torch.distributed.run
    (the parent has not touched the GPU and never will) and leave with their exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("[bench] WORLD_SIZE unset: launching %d ranks: %s" % (n, " ".join(cmd)))
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=64, help="frames per GPU per step")
    ap.add_argument("--chunk-frames", type=int, default=32, help="frames per kernel-launch chunk (0 = all frames of the step)")
    ap.add_argument("--lanes", type=int, default=2,
                    help="concurrent HIP streams, one chunk each.  Default: 2 streams x 32-frame chunks, the engine's throughput "
                         "configuration (stages of different chunks overlap; measured in round 3 against 4 x 16, 1 x 64, 4 x 32, "
                         "2 x 16, 4 x 8: 2 x 32 is 1-3 %% ahead of 4 x 16 and 6 %% ahead of one stream).  The kernel table and the "
                         "roofline block never come from these contended launches: they are timed in extra single-stream "
                         "passes after the timed region (see --single-stream)")
    ap.add_argument("--single-stream", action="store_true",
                    help="one stream, one chunk of all frames, in the timed region too: every kernel has the GPU to itself, so "
                         "rocprofv3 --kernel-trace of THIS command reports the durations of the kernel table "
                         "(profiles/rNN_kernel_stats_bench_single_stream.csv)")
    ap.add_argument("--sparse", action="store_true", help="depth off (hazard H20 extension)")
    ap.add_argument("--byte-masks", action="store_true",
                    help="instance masks as uint8 planes (V,M,H,W) -- the reference's np.uint8(mask) -- instead of one "
                         "bit-packed word per pixel (SURVEY.md 8d: 'M/8 mask-bits')")
    ap.add_argument("--boxes", type=int, nargs=2, default=[30, 40], help="objects per scene (min max)")
    ap.add_argument("--no-chain", action="store_true",
                    help="sequence the stage entry points from Python on every step (default: every chunk is ONE C call, "
                         "dfu3d_pseudo_boxes; same kernels, same results)")
    ap.add_argument("--graphs", action="store_true",
                    help="replay one captured hipGraph per chunk (measured slower than stream launches on ROCm 7.2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-procs", type=int, default=-1,
                    help="host processes of the all-cores CPU figure (-1 = every core this job may use: affinity mask cut "
                         "to the cgroup CPU quota; 0 = skip; forked before the GPU is initialised)")
    ap.add_argument("--rf-big-frames", type=int, default=192,
                    help="radius filter once more on the pool of this many frames in ONE launch (0 = skip): the default "
                         "step's shadow (92 / 184 MB) fits the 256 MiB memory-side cache it has just been written "
                         "through, a 192-frame pool (550 MB) does not -- the HBM-resident figure next to the cache-resident one")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--timing-steps", type=int, default=3,
                    help="extra passes AFTER the timed region in which every stage is bracketed with HIP events on the "
                         "launch stream (one stream, one chunk)")
    ap.add_argument("--dump", type=str, default="", help="write the JSON here as well")
    ap.add_argument("--dist-backend", default=None, help="nccl (default on GPUs) | gloo (rehearsal)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal: every rank uses cuda:0 (needs --dist-backend gloo)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))          # before anything touches the GPU

    cpu_pool = None
    under_profiler = ("rocprof" in os.environ.get("LD_PRELOAD", "").lower()
                      or any(k.upper().startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ))
    avail = usable_cores()
    n_cpu = avail if args.cpu_procs < 0 else min(args.cpu_procs, avail)
    if (int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cpu_baseline and n_cpu > 0
            and not torch.cuda.is_initialized() and not under_profiler):   # a profiler's preload already owns the GPU
        try:
            cpu_pool = cpu_all_cores_start(n_cpu)
        except Exception as e:                       # the single-thread baseline below does not depend on it
            log("all-cores CPU baseline disabled: %s" % e)

    if args.single_device:
        os.environ["LOCAL_RANK_OVERRIDE"] = "0"
    rank, world, local = D.init_from_env(args.dist_backend)
    if args.single_device:
        local = 0
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    if world > 1 and torch.distributed.get_world_size() != args.gpus:
        raise SystemExit("process group has %d ranks, expected %d" % (torch.distributed.get_world_size(), args.gpus))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    dev = torch.device("cuda", local if world > 1 else 0)
    torch.cuda.set_device(dev)
    dense = not args.sparse
    params = Params()
    frames = args.frames
    if args.single_stream:
        args.lanes, args.chunk_frames = 1, 0
    chunk_frames = min(args.chunk_frames or frames, frames)
    if frames % chunk_frames:
        raise SystemExit("--frames must be a multiple of --chunk-frames")

    t0 = time.time()
    # weak scaling: every rank labels its own `frames` frames; global frame ids are
    # interleaved over ranks exactly like dist.shard_frames(world*frames, rank, world)
    my_frames = D.shard_frames(world * frames, rank, world)
    scenes = [synth.make_scene(fid, H=H, W=W, M=MAX_INST, cams=CAMS, dense=dense, device=dev,
                               k_min=args.boxes[0], k_max=args.boxes[1]) for fid in my_frames]
    batch = synth.to_view_batch(scenes, params, dev, dense=dense, frame_ids=my_frames)
    mask_word = 0
    pack_ms = None
    if not args.byte_masks:
        # the resident input format: one word per pixel, bit j = instance j.  The reference hands uint8 planes
        # (my_loader.py:522-525): what converting them costs is measured here and reported next to `value`
        planes = batch.masks
        batch.pack_masks()
        mask_word = batch.mask_format
        if rank == 0:
            from dfu3d_amd import stages as _st
            V_, M_ = planes.shape[0], planes.shape[1]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                _st.pack_masks(planes.contiguous(), batch.n_inst.contiguous(), V_, M_, H, W, out=batch.masks, word_bytes=mask_word)
            e1.record()
            torch.cuda.synchronize()
            pack_ms = e0.elapsed_time(e1) / 3
        del planes
    cpu_scenes = None
    if rank == 0 and not args.no_cpu_baseline:
        keep = scenes[:10]
        cpu_scenes = [synth.Scene(s.points.cpu(), s.calibs, s.depth.cpu(), s.masks.cpu(), s.n_inst,
                                  s.inst_class, s.inst_box, s.inst_score, s.boxes3d) for s in keep]
    del scenes
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    log("[rank %d] %d frames generated in %.1fs" % (rank, frames, time.time() - t0))

    views = chunk_frames * CAMS

    def make_engine(lanes, vpc):
        return PseudoBoxEngine(params, H, W, MAX_INST, N_PTS, views_per_chunk=vpc, dense=dense, cap_vox=1 << 18,
                               pool_per_view=1 << 17, device=dev, lanes=lanes, graphs=args.graphs,
                               chain=not args.no_chain, free_lanes=True)

    eng = make_engine(args.lanes, views)
    gather_cap = eng.cap_rows * (frames * CAMS // views)      # rows a rank can emit per step: ONE fixed-size all-gather per step

    # Collecting, sorting and gathering the rows of pass i runs on a stream of its own that waits for pass i ONLY: on the launch
    # stream those small kernels and the host read would queue up behind pass i+1, the host would get the rows of pass i when
    # pass i+1 is over, and the GPU would run dry while the host sorts them and enqueues pass i+2.  Measured on one box, three runs
    # each (profiles/r04_collect_stream_sweep.log): 8 640-8 970 -> 9 480-9 660 frames/s.
    # The engine's lanes are FREE (no join between the lanes at the end of a pass: a lane's next chunk waits for that lane's previous
    # chunk only): 9 670-9 810 -> 9 830-9 960 frames/s on one box (profiles/r04_free_lanes_sweep.log); starting the second lane
    # half a chunk late on top of that: 9 890-10 020, inside the spread, not taken.
    side = torch.cuda.Stream(device=dev)

    def finish(handle):
        with torch.cuda.stream(side):
            rows, status = eng.collect(handle)
            if status:
                from dfu3d_amd.stages import status_message
                raise SystemExit("device status: " + status_message(status))
            out = D.allgather_rows(eng.gather_layout(rows, batch), cap_rows=gather_cap)
        side.synchronize()                           # (the rows are complete for whoever reads them next, on any stream)
        return out

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def steps(k):
        """k passes; pass i+1 is enqueued before the host waits for the rows of pass i (every pass is collected,
        sorted and gathered inside the loop -- nothing is left for after the timed region)."""
        rows, pending = None, None
        if args.graphs:                          # one set of captured row buffers: no pipelining across passes
            for _ in range(k):
                r_, st_ = eng.run(batch)
                if st_:
                    from dfu3d_amd.stages import status_message
                    raise SystemExit("device status: " + status_message(st_))
                rows = D.allgather_rows(eng.gather_layout(r_, batch), cap_rows=gather_cap)
            return rows
        for _ in range(k):
            h = eng.launch(batch)
            if pending is not None:
                rows = finish(pending)
            pending = h
        if pending is not None:
            before = rows
            rows = finish(pending)
            # the same 64 frames every pass: the rows of the last two passes must be the same rows (a check of ALL frames and
            # cameras that costs one comparison; the oracle comparison of the `parity` block covers ten frames)
            if before is not None:
                repeat["identical"] = bool(before.shape == rows.shape and torch.equal(before, rows))
        return rows

    repeat = {}
    rows = steps(args.warmup)
    barrier()
    t0 = time.perf_counter()
    rows = steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64,
                         device="cpu" if torch.distributed.get_backend() == "gloo" else dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    n_boxes = int(rows.shape[0])
    if world > 1:                              # every gathered row belongs to a frame of the global batch
        fr = rows[:, 0].long()
        assert int(fr.min()) >= 0 and int(fr.max()) < world * frames, "gathered rows carry foreign frame ids"
        assert torch.unique(fr % world).numel() == world or n_boxes < world, "rows of some rank are missing"

    # ---- kernel table: extra passes after the timed region, one stream, one chunk, every stage bracketed ----
    kern, counts, tsteps = {}, {}, 0
    if rank == 0 and not args.no_kernel_timing and args.timing_steps > 0:
        if args.lanes != 1 or chunk_frames != frames:
            del eng
            torch.cuda.empty_cache()
            eng = make_engine(1, frames * CAMS)
            eng.run(batch)
        tsteps = args.timing_steps
        eng.reset_timing()
        eng.timing = True
        for _ in range(tsteps):
            r_, st_ = eng.run(batch)
            eng._count("rows", r_.shape[0])
        eng.timing = False
        kern = eng.timing_summary()
        counts = eng.counters()
    # ---- radius filter on a pool larger than the memory-side cache (one launch over --rf-big-frames frames) ----
    rf_big = None
    if (rank == 0 and world == 1 and not args.no_kernel_timing and args.timing_steps > 0 and args.rf_big_frames > frames
            and dense):
        try:
            del eng
            torch.cuda.empty_cache()
            nb = args.rf_big_frames
            log("[bench] radius filter on %d frames in one launch ..." % nb)
            scenes_b = [synth.make_scene(fid, H=H, W=W, M=MAX_INST, cams=CAMS, dense=True, device=dev,
                                         k_min=args.boxes[0], k_max=args.boxes[1]) for fid in range(nb)]
            big = synth.to_view_batch(scenes_b, params, dev, dense=True, frame_ids=list(range(nb)))
            del scenes_b
            if not args.byte_masks:
                big.pack_masks()
            engb = PseudoBoxEngine(params, H, W, MAX_INST, N_PTS, views_per_chunk=nb * CAMS, dense=True, cap_vox=1 << 18,
                                   pool_per_view=1 << 17, device=dev, lanes=1, chain=False)
            engb.run(big)
            engb.reset_timing()
            engb.timing = True
            for _ in range(2):
                engb.run(big)
            engb.timing = False
            kb, cb = engb.timing_summary(), engb.counters()
            pts = cb["pool_points"] / 2
            ms = sum(kb[k][0] for k in RF_STAGE if k in kb) / 2
            rf_big = {"frames": nb, "points_per_launch": int(pts), "shadow_MB": round(16 * pts / 1e6, 1),
                      "avg_ms": round(ms, 4), "alg_bytes_per_launch": int(21 * pts),
                      "achieved": round(21 * pts / (ms * 1e-3) / 1e9, 2), "frac": round(21 * pts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                      "stages_ms": {k: round(kb[k][0] / 2, 4) for k in RF_STAGE if k in kb}}
            del engb, big
            torch.cuda.empty_cache()
        except Exception as e:                                       # never lose the timing line
            rf_big = {"error": repr(e)}
    copy_rate = measure_copy_rate(dev) if rank == 0 else None

    if rank == 0:
        total_frames = args.steps * frames * world
        value = total_frames / dt
        out = {
            "metric": "pseudo-box frames/sec (34k pts, 6 cams)", "value": round(value, 3),
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000.0 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1] with depth back-projection %s: %d synthetic frames/GPU/step, "
                                   "%d-pt sweep, %d cams x %dx%d depth + %d instance masks (%s), %d-%d objects/scene"
                                   % ("on (dense)" if dense else "off (sparse)", frames, N_PTS, CAMS, W, H, MAX_INST,
                                      "one bit-packed %d-byte word per pixel" % mask_word if mask_word else "uint8 planes",
                                      args.boxes[0], args.boxes[1]),
                       "frames_per_gpu_per_step": frames, "views_per_launch_chunk": views, "streams": args.lanes,
                       "boxes_per_step_all_ranks": n_boxes, "parallelism": "frames sharded x%d" % world},
            "hbm_copy_GBs_measured": round(copy_rate, 1),
        }
        if pack_ms is not None:
            # `value` is quoted on masks that are resident as packed words (SURVEY.md 8d prices "M/8 mask-bits"); the
            # reference's boundary hands uint8 planes: converting a step's planes, serialised in front of the step
            step_s = dt / args.steps
            out["mask_packing"] = {"pack_ms_per_step": round(pack_ms, 4), "bytes_read_per_step": int(frames * CAMS * MAX_INST * H * W),
                                   "value_pack_inclusive": round(frames * world / (step_s + pack_ms * 1e-3), 3),
                                   "how": "dfu3d_pack_masks on the step's (V, M, H, W) uint8 planes, HIP events, added to ms_per_step "
                                          "(not overlapped); `--byte-masks` runs the path on the planes themselves"}
        if kern:
            V1 = frames * CAMS                         # views per launch of the bracketed passes
            c1 = {k: v / tsteps for k, v in counts.items()}      # per pass
            for k in ("fov_points", "label_rows", "pool_points", "ball_points", "instance_points", "rows"):
                c1.setdefault(k, 0)
            alg = algorithmic_bytes(c1, V1, MAX_INST, mask_word)
            table = []
            for name, (ms, n) in sorted(kern.items(), key=lambda kv: -kv[1][0]):
                avg = ms / n
                row = {"stage": name, "kernels": STAGE_KERNELS.get(name, [name]), "total_ms": round(ms, 3),
                       "launches": n, "avg_ms": round(avg, 4)}
                if alg.get(name):
                    row["alg_bytes_per_launch"] = int(alg[name])
                    row["achieved_GBs"] = round(alg[name] / (avg * 1e-3) / 1e9, 2)
                    row["frac_of_peak"] = round(alg[name] / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
                table.append(row)
            out["kernels"] = table
            pass_ms = sum(ms for ms, _ in kern.values()) / tsteps
            pass_alg = sum(v for k, v in alg.items() if k != "rf_stage" and k in kern) + alg["rf_stage"]
            out["pass"] = {"alg_bytes": int(pass_alg), "ms": round(pass_ms, 3), "achieved": round(pass_alg / (pass_ms * 1e-3) / 1e9, 2),
                           "frac": round(pass_alg / (pass_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "unit": "GB/s",
                           "what": "algorithmic bytes of every stage of one single-stream pass over %d views / the sum of "
                                   "the bracketed stage durations" % V1}
            out["kernel_timing"] = {"how": "HIP events on the launch stream around every stage, %d passes after the timed "
                                           "region, one stream, %d views per launch" % (tsteps, V1),
                                    "sum_ms_per_pass": round(sum(ms for ms, _ in kern.values()) / tsteps, 3),
                                    "counts_per_pass": {k: int(v) for k, v in c1.items()}}
            dom = table[0]                              # the dominant kernel of the pass
            if "achieved_GBs" in dom:
                traffic, tsrc = pmc_traffic(dom["kernels"])
                out["roofline"] = {"bound": "hbm", "limited_by": valu_limit(dom["kernels"][0], dom["avg_ms"]),
                                   "kernel": dom["kernels"][0], "stage": dom["stage"],
                                   "achieved": dom["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": round(dom["achieved_GBs"] / HBM_PEAK_GBS, 5),
                                   "frac_of_measured_copy": round(dom["achieved_GBs"] / copy_rate, 5),
                                   "avg_ms": dom["avg_ms"], "alg_bytes_per_launch": dom["alg_bytes_per_launch"],
                                   "traffic": traffic, "traffic_source": tsrc,
                                   "timing": "single-stream passes after the timed region (%d views per launch); the same "
                                             "durations are in rocprofv3's summary of `python bench.py --single-stream` "
                                             "(profiles/rNN_kernel_stats_bench_single_stream.csv)" % V1}
            rf = [r for r in table if r["stage"] in RF_STAGE]
            if rf:
                ms_stage = sum(r["avg_ms"] for r in rf)
                gbs = alg["rf_stage"] / (ms_stage * 1e-3) / 1e9
                fl = next((r for r in rf if r["stage"] == "rf_flags"), None)
                traffic, tsrc = pmc_traffic([k for r in rf for k in r["kernels"]])
                out["radius_filter_roofline"] = {
                    "stage": {"kernels": [k for r in rf for k in r["kernels"]], "avg_ms": round(ms_stage, 4),
                              "alg_bytes_per_launch": int(alg["rf_stage"]), "achieved": round(gbs, 2),
                              "frac": round(gbs / HBM_PEAK_GBS, 5), "frac_of_measured_copy": round(gbs / copy_rate, 5),
                              "traffic": traffic},
                    "phase_A_stream_and_pair": None if fl is None else {
                        "avg_ms": fl["avg_ms"], "achieved": round(alg["rf_stage"] / (fl["avg_ms"] * 1e-3) / 1e9, 2),
                        "frac": round(alg["rf_stage"] / (fl["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "points_per_launch": int(c1["pool_points"]),
                    "undecided_after_phase_A": int(c1.get("rf_undecided", 0)),
                    "cache_note": "the shadow of this launch (%.0f MB) was written by k_seg_write just before and is read back "
                                  "through the 256 MiB memory-side cache; `big_pool` is the same stage on a pool that does not fit it"
                                  % (16 * c1["pool_points"] / 1e6),
                    "big_pool": rf_big}
        if cpu_scenes is not None and world == 1:
            fps, nf, secs, exp_rows = cpu_baseline(cpu_scenes, params, dense)
            try:
                out["parity"] = parity_block(rows, exp_rows, nf, my_frames)
            except Exception as e:                                   # never lose the timing line
                out["parity"] = {"error": repr(e)}
            if "identical" in repeat:
                out["parity"]["last_two_steps_identical"] = repeat["identical"]
            if not (out["parity"].get("rows_equal") and out["parity"].get("within_north_star_tolerance", False)
                    and out["parity"].get("last_two_steps_identical", True)):
                # a fast step with other rows than the reference's is not a result: say so where nobody can miss it
                log("[bench] PARITY FAILED: %s" % json.dumps(out["parity"]))
                out["parity_failed"] = True
            if cpu_pool is not None:
                try:
                    fps_all, t_all = cpu_all_cores_run(cpu_pool, cpu_scenes[0], params, dense)
                    out["cpu_baseline_allcores"] = {
                        "value": round(fps_all, 3), "unit": "frames/s", "cores": cpu_pool["n"], "kind": "port",
                        "sample": "%d processes = every core this job may use (affinity mask cut to the control group's CPU quota; "
                                  "the host has %d), each the same synthetic frame (6 cams) through the oracle at the same time; slowest %.1fs"
                                  % (cpu_pool["n"], os.cpu_count(), t_all)}
                    cpu_pool = None
                except Exception as e:
                    log("all-cores CPU baseline failed: %s" % e)
            out["cpu_baseline"] = {"value": round(fps, 4), "unit": "frames/s", "cores": 1, "kind": "port",
                                   "sample": "%d of the same synthetic frames (6 cams each) through oracle/penet_oracle.py "
                                             "(NumPy + C, 1 thread) in %.1fs; host has %d cores" % (nf, secs, os.cpu_count())}
        if cpu_pool is not None:
            cpu_all_cores_cancel(cpu_pool)
        if args.dump:
            with open(args.dump, "w") as f:
                json.dump(out, f, indent=1)
        print(json.dumps(out), flush=True)
    elif cpu_pool is not None:
        cpu_all_cores_cancel(cpu_pool)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
