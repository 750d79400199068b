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

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_COPY_GBS = 6290.0        # what a float4 device copy reaches on this part (MI355X_MICROARCH.md: 79 % of the spec peak)
H, W, CAMS, MAX_INST, N_PTS = 900, 1600, 6, 8, 34720


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_batch(frames, seed0, device, dense, k_boxes):
    scenes = [synth.make_scene(seed0 + f, H=H, W=W, M=MAX_INST, cams=CAMS, dense=dense,
                               device=device, k_min=k_boxes[0], k_max=k_boxes[1])
              for f in range(frames)]
    # masks/depth are produced on the device already; keep only what the engine needs
    return scenes


PMC_NAMES = {"bp_bin": "k_bp_bin", "bp_emit": "k_bp_emit", "rf_flags": "k_radius_flags",
             "fov_filter": "k_fov_filter", "project_label": "k_project_rows"}


def pmc_traffic(kernel, views_per_launch):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes
    (profiles/r01_pmc_traffic.json: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), scaled
    linearly when this run's launch covers a different number of views; None if unknown."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            pm = json.load(f)
        k = pm["kernels"][PMC_NAMES[kernel]]
        if k["fetch_bytes"] is None or k["write_bytes"] is None:
            return None
        return int((k["fetch_bytes"] + k["write_bytes"]) * views_per_launch / pm["views_per_launch"])
    except Exception:
        return None


def cpu_baseline(scenes_cpu, params, dense, max_seconds=18.0):
    """The oracle (NumPy + C restatement, single thread) on a bounded sample of
    the same workload, on this box's host cores."""
    from oracle import penet_oracle as O
    op = O.Params()
    t0 = time.time()
    done = 0
    for s in scenes_cpu:
        pts = s.points.numpy()
        for c, cal in enumerate(s.calibs):
            oc = O.Calibration({"P2": cal.P2, "R0": cal.R0, "Tr_velo2cam": cal.V2C})
            lid, _ = O.fov_filter(pts, oc, params.fov_hw)
            n = int(s.n_inst[c])
            d = s.depth[c].numpy().copy() if dense else np.zeros((H, W), np.float32)
            O.depth2pointsrgbpm(d[:, :, None], None, oc, lid, O.NUSC_CLASSES,
                                s.masks[c][:n].numpy().astype(np.float32),
                                s.inst_class[c][:n].numpy(), s.inst_box[c][:n].numpy(), op,
                                plane_key=c, want_points=False)
        done += 1
        if time.time() - t0 > max_seconds:
            break
    dt = time.time() - t0
    return done / dt, done, dt


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=64, help="frames per GPU per step")
    ap.add_argument("--chunk-frames", type=int, default=16, help="frames per kernel-launch chunk")
    ap.add_argument("--lanes", type=int, default=4, help="concurrent HIP streams (one chunk each)")
    ap.add_argument("--sparse", action="store_true", help="depth off (hazard H20 extension)")
    ap.add_argument("--boxes", type=int, nargs=2, default=[30, 40], help="objects per scene (min max)")
    ap.add_argument("--no-chain", action="store_true",
                    help="sequence the stage entry points from Python on every step (default: every chunk is ONE C call, "
                         "dfu3d_pseudo_boxes, except on the kernel-timing steps; same kernels, same results)")
    ap.add_argument("--graphs", action="store_true",
                    help="replay one captured hipGraph per chunk (measured slower than stream launches on ROCm 7.2: 3.7k vs 4.7k frames/s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-procs", type=int, default=16,
                    help="host processes of the all-cores CPU figure (0 = skip; forked before the GPU is initialised)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--single-stream", action="store_true",
                    help="after the timed region, time the same kernels once more with nothing else on the GPU (one stream, "
                         "all frames in one launch chain) and report them under \"single_stream\"")
    ap.add_argument("--timing-steps", type=int, default=3,
                    help="steps of the timed region whose kernels are bracketed with HIP events (the brackets "
                         "split multi-kernel stages into separate calls and cost a few percent)")
    ap.add_argument("--dump", type=str, default="", help="write per-kernel timing JSON here")
    ap.add_argument("--dist-backend", default=None, help="nccl (default on GPUs) | gloo (rehearsal)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal: every rank uses cuda:0 (needs --dist-backend gloo)")
    args = ap.parse_args()

    cpu_pool = None
    under_profiler = ("rocprof" in os.environ.get("LD_PRELOAD", "").lower()
                      or any(k.upper().startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ))
    if (int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cpu_baseline and args.cpu_procs > 0
            and not torch.cuda.is_initialized() and not under_profiler):   # a profiler's preload already owns the GPU
        try:
            cpu_pool = cpu_all_cores_start(min(args.cpu_procs, os.cpu_count() or 1))
        except Exception as e:                       # the single-thread baseline below does not depend on it
            log("all-cores CPU baseline disabled: %s" % e)

    if args.single_device:
        os.environ["LOCAL_RANK_OVERRIDE"] = "0"
    rank, world, local = D.init_from_env(args.dist_backend)
    if args.single_device:
        local = 0
    if world != args.gpus:
        log("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    dev = torch.device("cuda", local if world > 1 else 0)
    torch.cuda.set_device(dev)
    dense = not args.sparse
    params = Params()
    frames = args.frames
    if frames % args.chunk_frames:
        raise SystemExit("--frames must be a multiple of --chunk-frames")

    t0 = time.time()
    # weak scaling: every rank labels its own `frames` frames; global frame ids are
    # interleaved over ranks exactly like dist.shard_frames(world*frames, rank, world)
    my_frames = D.shard_frames(world * frames, rank, world)
    scenes = [synth.make_scene(fid, H=H, W=W, M=MAX_INST, cams=CAMS, dense=dense, device=dev,
                               k_min=args.boxes[0], k_max=args.boxes[1]) for fid in my_frames]
    batch = synth.to_view_batch(scenes, params, dev, dense=dense)
    cpu_scenes = None
    if rank == 0 and not args.no_cpu_baseline:
        keep = scenes[:10]
        cpu_scenes = [synth.Scene(s.points.cpu(), s.calibs, s.depth.cpu(), s.masks.cpu(), s.n_inst,
                                  s.inst_class, s.inst_box, s.inst_score, s.boxes3d) for s in keep]
    del scenes
    torch.cuda.synchronize()
    log("[rank %d] %d frames generated in %.1fs" % (rank, frames, time.time() - t0))

    eng = PseudoBoxEngine(params, H, W, MAX_INST, N_PTS, views_per_chunk=args.chunk_frames * CAMS,
                          dense=dense, cap_vox=1 << 18, pool_per_view=1 << 17, device=dev,
                          lanes=args.lanes, graphs=args.graphs, chain=not args.no_chain)

    def step():
        rows, status = eng.run(batch)
        if status:
            from dfu3d_amd.stages import status_message
            raise SystemExit("device status: " + status_message(status))
        return D.allgather_rows(rows)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        rows = step()
    eng.reset_timing()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        eng.timing = (not args.no_kernel_timing) and i < args.timing_steps
        rows = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64,
                         device="cpu" if torch.distributed.get_backend() == "gloo" else dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    n_boxes = int(rows.shape[0])

    kern = eng.timing_summary() if not args.no_kernel_timing else {}
    if rank == 0:
        total_frames = args.steps * frames * world
        value = total_frames / dt
        out = {
            "metric": "pseudo-box frames/sec (34k pts, 6 cams)", "value": round(value, 3),
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000.0 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1] with depth back-projection %s: %d synthetic frames/GPU/step, "
                                   "%d-pt sweep, %d cams x %dx%d depth + %d instance masks, %d-%d objects/scene"
                                   % ("on (dense)" if dense else "off (sparse)", frames, N_PTS, CAMS, W, H,
                                      MAX_INST, args.boxes[0], args.boxes[1]),
                       "frames_per_gpu_per_step": frames, "views_per_launch_chunk": args.chunk_frames * CAMS,
                       "streams": args.lanes,
                       "boxes_per_step_all_ranks": n_boxes, "parallelism": "frames sharded x%d" % world},
        }
        if kern:
            # algorithmic bytes per launch (DESIGN.md §kernels / SURVEY.md §8d)
            views_per_launch = args.chunk_frames * CAMS
            rf_pts = eng.rf_points_total()
            n_rf_launch = kern.get("rf_flags", (0, 1))[1]
            alg = {
                "bp_bin": views_per_launch * H * W * 4,                  # depth read
                "bp_emit": views_per_launch * H * W * 4,                 # bin-id read
                "rf_flags": 21.0 * rf_pts / max(n_rf_launch, 1),          # 16n + 4n + 1n
                "fov_filter": views_per_launch * N_PTS * 16,
                "project_label": views_per_launch * N_PTS * 16,
            }
            table = []
            for name, (ms, n) in sorted(kern.items(), key=lambda kv: -kv[1][0]):
                avg = ms / n
                row = {"kernel": name, "total_ms": round(ms, 3), "launches": n, "avg_ms": round(avg, 4)}
                if name in alg:
                    row["alg_bytes_per_launch"] = int(alg[name])
                    row["achieved_GBs"] = round(alg[name] / (avg * 1e-3) / 1e9, 2)
                table.append(row)
            out["kernels"] = table
            dom = next((r for r in table if "achieved_GBs" in r), None)
            if dom is not None:
                out["roofline"] = {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["achieved_GBs"],
                                   "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": round(dom["achieved_GBs"] / HBM_PEAK_GBS, 5),
                                   "frac_of_copy_peak": round(dom["achieved_GBs"] / HBM_COPY_GBS, 5),
                                   "traffic": pmc_traffic(dom["kernel"], views_per_launch)}
            rf = next((r for r in table if r["kernel"] == "rf_flags"), None)
            if rf is not None:
                out["radius_filter_roofline"] = {"achieved": rf["achieved_GBs"], "peak": HBM_PEAK_GBS,
                                                 "unit": "GB/s", "frac": round(rf["achieved_GBs"] / HBM_PEAK_GBS, 6),
                                                 "frac_of_copy_peak": round(rf["achieved_GBs"] / HBM_COPY_GBS, 6),
                                                 "points_per_launch": int(rf_pts / max(n_rf_launch, 1))}
        if kern and world == 1 and args.single_stream:
            # the same kernels with nothing else on the GPU: one stream, all frames in one launch chain.
            # (Informational, outside the timed region: with several streams the durations above include the
            # contention that buys the overlap of the single-workgroup tails.)
            del eng
            torch.cuda.empty_cache()
            eng1 = PseudoBoxEngine(params, H, W, MAX_INST, N_PTS, views_per_chunk=frames * CAMS, dense=dense,
                                   cap_vox=1 << 18, pool_per_view=1 << 17, device=dev, lanes=1)
            eng1.run(batch)
            eng1.timing = True
            eng1.reset_timing()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                eng1.run(batch)
            torch.cuda.synchronize()
            ms1 = (time.perf_counter() - t1) / 3 * 1e3
            k1 = eng1.timing_summary()
            v1 = frames * CAMS
            pts1 = eng1.rf_points_total() / max(k1.get("rf_flags", (0, 1))[1], 1)
            ss = {"ms_per_step": round(ms1, 3), "frames_per_s": round(frames / (ms1 * 1e-3), 1)}
            for name, byts in (("bp_bin", v1 * H * W * 4), ("bp_emit", v1 * H * W * 4), ("rf_flags", 21.0 * pts1)):
                if name in k1:
                    avg = k1[name][0] / k1[name][1]
                    ss[name] = {"avg_ms": round(avg, 4), "achieved_GBs": round(byts / (avg * 1e-3) / 1e9, 1),
                                "frac": round(byts / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            out["single_stream"] = ss
            del eng1
        if cpu_scenes is not None and world == 1:
            fps, nf, secs = cpu_baseline(cpu_scenes, params, dense)
            if cpu_pool is not None:
                try:
                    fps_all, t_all = cpu_all_cores_run(cpu_pool, cpu_scenes[0], params, dense)
                    out["cpu_baseline_all_cores"] = {
                        "value": round(fps_all, 3), "unit": "frames/s", "cores": cpu_pool["n"], "kind": "port",
                        "sample": "%d processes, each the same synthetic frame (6 cams) through the oracle at the same time; "
                                  "slowest %.1fs" % (cpu_pool["n"], t_all)}
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
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
