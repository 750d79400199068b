"""CLI mirror of `python main.py --command evaluate ...`
(tools/PENet/main.py:38-180, README.md:75-78).

Keeps the reference's flags (model flags are accepted and ignored: SEEM and
DepthAnything are input providers outside this path) and adds
--seg-dir/--depth-dir for precomputed inputs, --label-out, --start/--end and
--skip-existing.  Frames are the sorted stems of <detpath>/velodyne
(my_loader_KittiPandasetWaymo.py:1083-1090); with torchrun every rank takes
frames rank, rank+world, ... and writes its own label files."""
import argparse
import json
import os
import sys
import time

import numpy as np


def build_parser():
    ap = argparse.ArgumentParser(description='DFU3D pseudo-box generation (MI355X)')
    ap.add_argument('--command', default="evaluate")
    ap.add_argument('-n', '--network-model', type=str, default="pe", choices=["e", "pe"])
    ap.add_argument('--workers', default=4, type=int)
    ap.add_argument('--detpath', default='../../data/kitti/training', type=str)
    ap.add_argument('-e', '--evaluate', default='pe.pth.tar', type=str)
    ap.add_argument('--test', action="store_true", default=True)
    ap.add_argument('--cpu', action="store_true", default=False,
                    help='the reference\'s flag; this build has no CPU path and refuses it')
    ap.add_argument('--model', type=str, default='zoedepth')
    ap.add_argument('--pretrained_resource', type=str, default='')
    ap.add_argument('--conf_files', nargs='+', default=[])
    ap.add_argument('--user_dir', default=None)
    ap.add_argument('--config_overrides', nargs='*')
    ap.add_argument('--overrides', nargs=argparse.REMAINDER, default=[])
    # additive
    ap.add_argument('--depth-dir', default=None, help='<dir>/<idx>.npy dense metric depth (H,W); default <detpath>/depth_2')
    ap.add_argument('--seg-dir', default=None, help='<dir>/<idx>.npz instance masks / classes / scores / boxes; default <detpath>/seg_2')
    ap.add_argument('--label-out', default=None, help='default <detpath>/label_2')
    ap.add_argument('--start', type=int, default=0)
    ap.add_argument('--end', type=int, default=None)
    ap.add_argument('--skip-existing', action='store_true', help='restart: skip frames whose label file exists')
    ap.add_argument('--resume', action='store_true',
                    help='restart from the per-frame logs: skip frames that a frames.rank*.jsonl of an earlier run lists '
                         'as done (any rank count), and those whose label file exists')
    ap.add_argument('--keep-going', action='store_true',
                    help='one-frame-at-a-time path: a frame whose inputs cannot be read is logged (frames.rank*.jsonl, '
                         '"error") and skipped instead of ending the run')
    ap.add_argument('--batch-frames', type=int, default=16,
                    help='frames per engine call (0: one view at a time through save_depth_as_points)')
    ap.add_argument('--no-virtual-points', action='store_true',
                    help='labels only: skip the velodyne_depth/*.npy files of vis_utils.py:164-166')
    ap.add_argument('--streams', type=int, default=2)
    ap.add_argument('--reader-procs', type=int, default=8,
                    help='forked reader processes for the batched path (0 = reader threads only)')
    ap.add_argument('--dist-backend', default=None, help='nccl (default on GPUs: RCCL) | gloo (rehearsal)')
    ap.add_argument('--single-device', action='store_true',
                    help='rehearsal of a multi-rank run on one GPU: every rank uses cuda:0 (needs --dist-backend gloo)')
    return ap


def frame_list(detpath):
    return sorted(f[:-4] for f in os.listdir(os.path.join(detpath, 'velodyne')) if f.endswith('.bin'))


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.cpu:
        raise SystemExit("--cpu: this build has no CPU path (the oracle under oracle/ is test-only)")
    # reader processes must be forked before this process touches the GPU (dfu3d_amd/reader_pool.py)
    rpool = None
    if args.batch_frames > 0 and args.reader_procs > 0:
        try:
            from ..reader_pool import ReaderPool
            rpool = ReaderPool(args.reader_procs)
        except RuntimeError:
            rpool = None                      # GPU already initialised in this process: threads
    from .. import dist as D
    from .vis_utils import save_depth_as_points
    rank, world, local = D.init_from_env(args.dist_backend)
    if args.single_device:
        local = 0
    frames = frame_list(args.detpath)[args.start:args.end]
    mine = [frames[i] for i in D.shard_frames(len(frames), rank, world)]
    depth_dir = args.depth_dir or os.path.join(args.detpath, 'depth_2')
    label_out = args.label_out or os.path.join(args.detpath, 'label_2')
    t0 = time.time()
    n_assigned = len(mine)
    if args.resume:
        args.skip_existing = True
    if args.skip_existing:
        done = set()
        if args.resume and os.path.isdir(label_out):
            for fn in os.listdir(label_out):                    # per-frame logs of earlier runs (any rank count)
                if fn.startswith('frames.rank') and fn.endswith('.jsonl'):
                    with open(os.path.join(label_out, fn)) as f:
                        for line in f:
                            try:
                                rec = json.loads(line)
                            except ValueError:
                                continue                         # a line cut off by the crash we are restarting from
                            if 'error' not in rec:
                                done.add(rec.get('frame'))
        mine = [s for s in mine if s not in done and not os.path.exists(os.path.join(label_out, s + '.txt'))]
    errors = []
    frame_index = {s: i for i, s in enumerate(frames)}     # global index of a frame = its place in the sorted list
    rows_log = []                                           # (stem, engine rows) of every frame this rank labelled
    # per-frame log of this rank (JSONL per frame; also the done-list --resume reads): one line
    # per frame, appended and flushed as soon as the frame's label file is on disk, so a crashed run leaves its log
    import threading
    os.makedirs(label_out, exist_ok=True)
    log_f = open(os.path.join(label_out, 'frames.rank%d.jsonl' % rank), 'a')
    log_lock = threading.Lock()

    def log_frame(stem, r=None, error=None):
        rec = {"frame": stem, "frame_idx": frame_index[stem], "rank": rank}
        if error is None:
            rec["boxes"] = int(np.asarray(r).reshape(-1, 24).shape[0])
        else:
            rec["error"] = error
        with log_lock:
            log_f.write(json.dumps(rec) + "\n")
            log_f.flush()
    if args.batch_frames > 0:
        from ..pipeline import BatchedLabeler
        lab = BatchedLabeler(batch_frames=max(1, min(args.batch_frames, len(mine))), lanes=args.streams,
                             workers=max(1, args.workers),
                             device="cuda:%d" % (local if world > 1 else 0),
                             want_points=not args.no_virtual_points, reader_pool=rpool)
        lab.on_frame = log_frame
        stats = lab.run(args.detpath, mine, label_out, depth_dir, seg_dir=args.seg_dir)
        rows_log += lab.rows_log
        lab.close()
        if rank == 0:
            dt = time.time() - t0
            print("%d frames, %d boxes, %.2f frames/s (files in, files out)" % (stats["frames"], stats["boxes"],
                                                                              stats["frames"] / max(dt, 1e-9)))
        mine = []
    for k, idx in enumerate(mine):
        if args.skip_existing and os.path.exists(os.path.join(label_out, idx + '.txt')):
            continue
        from .vis_utils import load_seg_npz
        try:
            depth = np.load(os.path.join(depth_dir, idx + '.npy')).astype(np.float32)
            _, r = save_depth_as_points(depth, idx, args.detpath, label_root=label_out,
                                        seg_provider=(lambda path, _i=idx: load_seg_npz(args.detpath, _i, args.seg_dir)),
                                        device="cuda:%d" % (local if world > 1 else 0), return_rows=True)
        except (OSError, ValueError, KeyError) as e:             # unreadable / malformed input files of this frame
            if not args.keep_going:
                raise
            errors.append((idx, "%s: %s" % (type(e).__name__, e)))
            continue
        rows_log.append((idx, r))
        log_frame(idx, r)
        if rank == 0 and (k + 1) % 10 == 0:
            print("%d/%d frames, %.2f frames/s" % (k + 1, len(mine), (k + 1) / (time.time() - t0)))
    if rpool is not None:
        rpool.close()
    for stem, msg in errors:                                # (after the labelled frames, as before)
        log_frame(stem, error=msg)
    log_f.close()
    # the one collective of the path: variable-length all-gather of the box rows, each carrying its
    # global frame index; rank 0 writes the manifest of the run next to the label files
    import torch
    dev = torch.device("cuda:%d" % (local if world > 1 else 0)) if torch.cuda.is_available() else torch.device("cpu")
    parts = []
    for stem, r in rows_log:
        r = np.asarray(r, np.float64).reshape(-1, 24)
        if r.shape[0]:
            loc = torch.as_tensor(r, device=dev).clone()
            loc[:, 0] = 0
            parts.append(D.rows_for_gather(loc, [0], [frame_index[stem]], [0]))
    mine_rows = torch.cat(parts, 0) if parts else torch.zeros((0, len(D.GATHER_COLUMNS)), dtype=torch.float64, device=dev)
    gathered = D.allgather_rows(mine_rows)
    if world > 1:
        import torch.distributed as dist
        cnt = torch.tensor([len(rows_log), n_assigned], dtype=torch.int64,
                           device=dev if dist.get_backend() != "gloo" else "cpu")
        dist.all_reduce(cnt)
        n_labelled, n_total = int(cnt[0]), int(cnt[1])
    else:
        n_labelled, n_total = len(rows_log), n_assigned
    if rank == 0:
        os.makedirs(label_out, exist_ok=True)
        man = D.write_manifest(os.path.join(label_out, "manifest.json"), gathered, n_total, world,
                               extra={"frames_labelled_this_run": n_labelled, "partial": bool(n_labelled < n_total),   # (a --resume / --skip-existing run lists only its own boxes)
                                      "seconds": round(time.time() - t0, 3),
                                      "detpath": os.path.abspath(args.detpath)})
        print("manifest: %d frames (%d labelled now), %d boxes, per rank %s" % (
            man["frames"], n_labelled, man["boxes"], man["boxes_per_rank"]))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    return 0


if __name__ == "__main__":
    sys.exit(main())
