"""Drop-in for pcdet/datasets/kitti/kitti_object_eval_python/eval.py on the MI355X: the KITTI AP evaluator behind
`KittiDataset.evaluation` (kitti_dataset.py:421-431) -- `get_official_eval_result`, `do_eval`, `eval_class`,
`calculate_iou_partly`, `get_mAP`, `get_mAP_R40`, `get_thresholds`, same arguments and return values.

The reference runs numba-jitted CPU loops per frame and score threshold around a numba-CUDA IoU kernel that has no ROCm
target, and re-derives the ignore flags of every box in Python for every (class, difficulty).  Here the whole split is
one batch on the GPU (csrc/eval_stage.hip): one launch computes the per-frame overlap blocks of a metric, one launch
the matched scores of EVERY (class, difficulty, min_overlap) cell, the recall thresholds are picked on the host
(`get_thresholds`, a few hundred numbers), and one more launch counts tp / fp / fn for every cell, threshold and frame.
There is no CPU path: without libdfu3d_hip.so every function raises.
"""
import io

import numpy as np
import torch

from .. import stages as st

CLASS_NAMES = ['Car', 'Truck', 'Construction_vehicle', 'Bus', 'Trailer', 'Barrier', 'Motorcycle', 'Bicycle',
               'Pedestrian', 'Traffic_cone']                                 # eval.py:32 (the nuScenes2KITTI classes)
N_SAMPLE_PTS = 41
_LOWER = [n.lower() for n in CLASS_NAMES]
# names that are neutral -- neither a hit nor a miss -- for a class (eval.py:47-51)
_NEUTRAL = {'van': 'car', 'person_sitting': 'pedestrian'}


def _code(name):
    n = str(name).lower()
    if n in _LOWER:
        return _LOWER.index(n)
    if n in _NEUTRAL and _NEUTRAL[n] in _LOWER:
        return 1000 + _LOWER.index(_NEUTRAL[n])
    return -1


def _cat(annos, key, width, dtype=np.float64, optional=False):
    shape = (-1, width) if width else (-1,)
    parts = [(np.zeros((len(a['name']),) + shape[1:], dtype) if optional and key not in a else
              np.asarray(a[key], dtype).reshape(shape)) for a in annos]
    if not parts:
        return np.zeros((0, width) if width else (0,), dtype)
    return np.concatenate(parts, 0)


def pack_frames(gt_annos, dt_annos, device="cuda:0"):
    """Annotation lists (kitti_common.get_label_annos / generate_prediction_dicts) -> stages.EvalFrames on the GPU."""
    if len(gt_annos) != len(dt_annos):
        raise ValueError("as many ground-truth frames as detection frames are needed")
    if len(gt_annos) == 0:
        raise ValueError("no frames")
    G = np.array([len(a['name']) for a in gt_annos], np.int64)
    D = np.array([len(a['name']) for a in dt_annos], np.int64)
    if D.max(initial=0) > st.EVAL_MAX_DET:
        raise st.Dfu3dError("a frame has %d detections; the evaluator takes at most %d" % (D.max(), st.EVAL_MAX_DET))
    off = lambda n: np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
    names_g = [str(n) for a in gt_annos for n in a['name']]
    names_d = [str(n) for a in dt_annos for n in a['name']]
    cam = lambda annos: np.concatenate([_cat(annos, 'location', 3), _cat(annos, 'dimensions', 3),
                                        _cat(annos, 'rotation_y', 0)[:, None]], 1)
    host = dict(
        gt_off=off(G), dt_off=off(D), ov_off=off(G * D),
        gt_code=np.array([_code(n) for n in names_g], np.int32),
        gt_dontcare=np.array([n == 'DontCare' for n in names_g], np.int32),
        gt_bbox=_cat(gt_annos, 'bbox', 4), gt_alpha=_cat(gt_annos, 'alpha', 0),
        gt_occluded=_cat(gt_annos, 'occluded', 0, optional=True).astype(np.int32),
        gt_truncated=_cat(gt_annos, 'truncated', 0, optional=True),
        gt_cam=cam(gt_annos),
        dt_code=np.array([_code(n) for n in names_d], np.int32),
        dt_bbox=_cat(dt_annos, 'bbox', 4), dt_alpha=_cat(dt_annos, 'alpha', 0), dt_score=_cat(dt_annos, 'score', 0, optional=True),
        dt_cam=cam(dt_annos))
    # a detection is a hit only for its own class: neutral names have no meaning on that side (eval.py:75-79)
    host['dt_code'] = np.where(host['dt_code'] >= 1000, -1, host['dt_code']).astype(np.int32)
    dev = torch.device(device)
    fr = st.EvalFrames(**{k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in host.items()},
                       n_pairs=int((G * D).sum()), max_dt=int(D.max(initial=0)))
    fr.gt_counts, fr.dt_counts = G, D
    return fr


def get_thresholds(scores, num_gt, num_sample_pts=N_SAMPLE_PTS):
    """eval.py:9-27: walk the scores from the best down and keep the one at which the recall reaches the next of the
    `num_sample_pts` sample points.  Same arithmetic as the reference; instead of testing score after score, the first
    score that is kept for the current sample point is found with one vectorised comparison."""
    s = np.sort(np.asarray(scores, np.float64))[::-1]
    n = len(s)
    if n == 0:
        return []
    idx = np.arange(n, dtype=np.int64)
    left = (idx + 1) / num_gt
    right = np.where(idx < n - 1, (idx + 2) / num_gt, left)
    out, current, pos = [], 0.0, 0
    step = 1 / (num_sample_pts - 1.0)
    while pos < n:
        skip = ((right[pos:] - current) < (current - left[pos:])) & (idx[pos:] < n - 1)
        take = pos + int(np.argmin(skip))            # the last score is never skipped, so there is a False
        out.append(float(s[take]))
        current += step
        pos = take + 1
    return out


class _Evaluator:
    """The frames of one evaluation on the GPU, overlaps cached per metric."""

    def __init__(self, gt_annos, dt_annos, device="cuda:0"):
        self.fr = pack_frames(gt_annos, dt_annos, device)
        self._ov = {}

    def overlaps(self, metric):
        if metric not in (0, 1, 2):
            raise ValueError("unknown metric")
        if metric not in self._ov:
            self._ov[metric] = st.eval_overlaps(self.fr, metric)
        return self._ov[metric]

    def eval_class(self, current_classes, difficultys, metric, min_overlaps, compute_aos=False):
        fr = self.fr
        ov = self.overlaps(metric)
        min_overlaps = np.asarray(min_overlaps, np.float64)
        cells = [(m, l, k) for m in range(len(current_classes)) for l in range(len(difficultys))
                 for k in range(min_overlaps.shape[0])]
        combos = [(int(current_classes[m]), int(difficultys[l]), float(min_overlaps[k, metric, m])) for m, l, k in cells]
        matched, n_valid = st.eval_match_scores(fr, ov, metric, combos)
        matched, total_valid = matched.cpu().numpy(), n_valid.sum(dim=1).cpu().numpy()
        thresholds = np.zeros((len(cells), N_SAMPLE_PTS))
        n_thresh = np.zeros(len(cells), np.int32)
        for c in range(len(cells)):
            row = matched[c]
            th = get_thresholds(row[~np.isnan(row)], int(total_valid[c]))
            if len(th) > thresholds.shape[1]:
                thresholds = np.pad(thresholds, ((0, 0), (0, len(th) - thresholds.shape[1])))
            thresholds[c, :len(th)] = th
            n_thresh[c] = len(th)
        dev = fr.gt_off.device
        pr, sim = st.eval_match_stats(fr, ov, metric, combos, torch.from_numpy(thresholds).to(dev),
                                      torch.from_numpy(n_thresh).to(dev), compute_aos)
        pr = pr.cpu().numpy().astype(np.float64)
        sim = sim.sum(dim=1).cpu().numpy() if sim is not None else None
        shape = [len(current_classes), len(difficultys), min_overlaps.shape[0], N_SAMPLE_PTS]
        precision, recall, aos = np.zeros(shape), np.zeros(shape), np.zeros(shape)
        with np.errstate(invalid='ignore', divide='ignore'):
            for c, (m, l, k) in enumerate(cells):
                n = min(int(n_thresh[c]), N_SAMPLE_PTS)
                tp, fp, fn = pr[c, :n, 0], pr[c, :n, 1], pr[c, :n, 2]
                recall[m, l, k, :n] = tp / (tp + fn)
                precision[m, l, k, :n] = tp / (tp + fp)
                if compute_aos:
                    aos[m, l, k, :n] = sim[c, :n] / (tp + fp)
                # eval.py:563-570: every sample point takes the best value at or behind it -- over all 41 columns
                for arr in ((precision, recall, aos) if compute_aos else (precision, recall)):
                    for i in range(n):
                        arr[m, l, k, i] = np.max(arr[m, l, k, i:], axis=-1)
        return {"recall": recall, "precision": precision, "orientation": aos}

    def frame_overlaps(self, metric):
        """-> list over frames of (detections, ground truths) float64 matrices (the `overlaps` of eval.py:483-484)."""
        fr, ov = self.fr, self.overlaps(metric).cpu().numpy()
        off = fr.ov_off.cpu().numpy()
        return [ov[off[f]:off[f + 1]].reshape(int(fr.gt_counts[f]), int(fr.dt_counts[f])).T.copy() for f in range(fr.F)]


def eval_class(gt_annos, dt_annos, current_classes, difficultys, metric, min_overlaps, compute_aos=False,
               num_parts=100, _evaluator=None):
    """eval.py:447-577.  min_overlaps: (num_overlap, metric, class).  `num_parts` is accepted and ignored: it only
    bounds the reference's dense cross-frame matrices.  -> {'recall', 'precision', 'orientation'}, each
    (class, difficulty, num_overlap, 41)."""
    ev = _evaluator or _Evaluator(gt_annos, dt_annos)
    return ev.eval_class(current_classes, difficultys, metric, min_overlaps, compute_aos)


def calculate_iou_partly(gt_annos, dt_annos, metric, num_parts=50):
    """eval.py:336-417 -> (overlaps, parted_overlaps, total_gt_num, total_dt_num) with overlaps[f] of shape
    (len(gt_annos[f]), len(dt_annos[f])).  parted_overlaps -- the reference's dense cross-frame blocks -- is None."""
    ev = _Evaluator(dt_annos, gt_annos)             # frame_overlaps() puts its `dt` side first
    total_gt = np.array([len(a['name']) for a in gt_annos], np.int64)
    total_dt = np.array([len(a['name']) for a in dt_annos], np.int64)
    return ev.frame_overlaps(metric), None, total_gt, total_dt


def get_mAP(prec):
    """eval.py:580-584: 11-point interpolation (summed in the reference's order)."""
    total = 0
    for i in range(0, prec.shape[-1], 4):
        total = total + prec[..., i]
    return total / 11 * 100


def get_mAP_R40(prec):
    """eval.py:587-591: 40 recall positions."""
    total = 0
    for i in range(1, prec.shape[-1]):
        total = total + prec[..., i]
    return total / 40 * 100


def do_eval(gt_annos, dt_annos, current_classes, min_overlaps, compute_aos=False, difficultys=(0, 1, 2),
            PR_detail_dict=None):
    """eval.py:645-690 -> (mAP_bbox, mAP_bev, mAP_3d, mAP_aos, and the four R40 tables)."""
    ev = _Evaluator(gt_annos, dt_annos)
    ret = ev.eval_class(current_classes, difficultys, 0, min_overlaps, compute_aos)
    bbox, bbox40 = get_mAP(ret["precision"]), get_mAP_R40(ret["precision"])
    if PR_detail_dict is not None:
        PR_detail_dict['bbox'] = ret['precision']
    aos = aos40 = None
    if compute_aos:
        aos, aos40 = get_mAP(ret["orientation"]), get_mAP_R40(ret["orientation"])
        if PR_detail_dict is not None:
            PR_detail_dict['aos'] = ret['orientation']
    ret = ev.eval_class(current_classes, difficultys, 1, min_overlaps)
    bev, bev40 = get_mAP(ret["precision"]), get_mAP_R40(ret["precision"])
    if PR_detail_dict is not None:
        PR_detail_dict['bev'] = ret['precision']
        PR_detail_dict['bev_recall'] = ret['recall']
    ret = ev.eval_class(current_classes, difficultys, 2, min_overlaps)
    d3, d340 = get_mAP(ret["precision"]), get_mAP_R40(ret["precision"])
    if PR_detail_dict is not None:
        PR_detail_dict['3d'] = ret['precision']
        PR_detail_dict['3d_recall'] = ret['recall']
    return bbox, bev, d3, aos, bbox40, bev40, d340, aos40


def official_min_overlaps():
    """eval.py:719-733: the reference evaluates every class at 0.5 (twice) and at 0.25 -- (3, metric, 10 classes)."""
    return np.stack([np.full((3, 10), 0.5), np.full((3, 10), 0.5), np.full((3, 10), 0.25)], axis=0)


def get_official_eval_result(gt_annos, dt_annos, current_classes, PR_detail_dict=None):
    """eval.py:708-845 -> (report text, {'<Class>_<3d|bev|image|aos>/<easy|moderate|hard>_R40': AP})."""
    if not isinstance(current_classes, (list, tuple)):
        current_classes = [current_classes]
    classes = [CLASS_NAMES.index(c) if isinstance(c, str) else int(c) for c in current_classes]
    min_overlaps = official_min_overlaps()[:, :, classes]
    compute_aos = False
    for anno in dt_annos:                            # alpha == -10 marks "no orientation" (eval.py:768-773)
        if anno['alpha'].shape[0] != 0:
            compute_aos = bool(anno['alpha'][0] != -10)
            break
    bbox, bev, d3, aos, bbox40, bev40, d340, aos40 = do_eval(gt_annos, dt_annos, classes, min_overlaps, compute_aos,
                                                             PR_detail_dict=PR_detail_dict)
    out = io.StringIO()
    ret = {}
    levels = ('easy', 'moderate', 'hard')

    def table(tag, t, j, i, digits):
        print("%s AP:%s" % (tag, ", ".join("%.*f" % (digits, t[j, l, i]) for l in range(3))), file=out)

    for j, cls in enumerate(classes):
        name = CLASS_NAMES[cls]
        for i in range(min_overlaps.shape[0]):
            head = "{:.2f}, {:.2f}, {:.2f}:".format(*min_overlaps[i, :, j])
            print("%s AP@%s" % (name, head), file=out)
            table("bbox", bbox, j, i, 4)
            table("bev ", bev, j, i, 4)
            table("3d  ", d3, j, i, 4)
            if compute_aos:
                table("aos ", aos, j, i, 2)
            print("%s AP_R40@%s" % (name, head), file=out)
            table("bbox", bbox40, j, i, 4)
            table("bev ", bev40, j, i, 4)
            table("3d  ", d340, j, i, 4)
            if compute_aos:
                table("aos ", aos40, j, i, 2)
            if i == 0:
                kinds = ([('aos', aos40)] if compute_aos else []) + [('3d', d340), ('bev', bev40), ('image', bbox40)]
                for kind, t in kinds:
                    for l, level in enumerate(levels):
                        ret['%s_%s/%s_R40' % (name, kind, level)] = t[j, l, 0]
    return out.getvalue(), ret


def do_coco_style_eval(gt_annos, dt_annos, current_classes, overlap_ranges, compute_aos):
    """eval.py:675-690: AP averaged over ten min-overlaps per (metric, class).  overlap_ranges (3, metric, class) =
    (first, last, count) for np.linspace.  The reference's version cannot run: it hands np.linspace a float count
    (TypeError under any recent NumPy) and unpacks FOUR values from its own do_eval, which returns eight
    (eval.py:645-673); golden G9 records the exception.  This one does what the statements mean."""
    min_overlaps = np.zeros([10, *overlap_ranges.shape[1:]])
    for i in range(overlap_ranges.shape[1]):
        for j in range(overlap_ranges.shape[2]):
            lo, hi, num = overlap_ranges[:, i, j]
            min_overlaps[:, i, j] = np.linspace(lo, hi, int(num))
    mAP_bbox, mAP_bev, mAP_3d, mAP_aos = do_eval(gt_annos, dt_annos, current_classes, min_overlaps, compute_aos)[:4]
    mAP_bbox, mAP_bev, mAP_3d = mAP_bbox.mean(-1), mAP_bev.mean(-1), mAP_3d.mean(-1)
    if mAP_aos is not None:
        mAP_aos = mAP_aos.mean(-1)
    return mAP_bbox, mAP_bev, mAP_3d, mAP_aos


def _compute_aos(dt_annos):
    for anno in dt_annos:                            # alpha == -10 marks "no orientation" (eval.py:768-773)
        if anno['alpha'].shape[0] != 0:
            return bool(anno['alpha'][0] != -10)
    return False


def get_coco_eval_result(gt_annos, dt_annos, current_classes):
    """eval.py:836-894 -> report text.  The class table is the function's own (KITTI's five names by number), as in
    the reference: a class NUMBER handed to do_eval still indexes the module's CLASS_NAMES (eval.py:32)."""
    class_to_name = {0: 'Car', 1: 'Pedestrian', 2: 'Cyclist', 3: 'Van', 4: 'Person_sitting'}
    class_to_range = {0: [0.5, 0.95, 10], 1: [0.25, 0.7, 10], 2: [0.25, 0.7, 10], 3: [0.5, 0.95, 10], 4: [0.25, 0.7, 10]}
    name_to_class = {v: n for n, v in class_to_name.items()}
    if not isinstance(current_classes, (list, tuple)):
        current_classes = [current_classes]
    current_classes = [name_to_class[c] if isinstance(c, str) else c for c in current_classes]
    overlap_ranges = np.zeros([3, 3, len(current_classes)])
    for i, curcls in enumerate(current_classes):
        overlap_ranges[:, :, i] = np.array(class_to_range[curcls])[:, np.newaxis]
    compute_aos = _compute_aos(dt_annos)
    mAPbbox, mAPbev, mAP3d, mAPaos = do_coco_style_eval(gt_annos, dt_annos, current_classes, overlap_ranges, compute_aos)
    out = io.StringIO()
    for j, curcls in enumerate(current_classes):
        o_range = np.array(class_to_range[curcls])[[0, 2, 1]]
        o_range[1] = (o_range[2] - o_range[0]) / (o_range[1] - 1)
        print("%s coco AP@%.2f:%.2f:%.2f:" % ((class_to_name[curcls],) + tuple(o_range)), file=out)
        print("bbox AP:%.2f, %.2f, %.2f" % tuple(mAPbbox[j, :3]), file=out)
        print("bev  AP:%.2f, %.2f, %.2f" % tuple(mAPbev[j, :3]), file=out)
        print("3d   AP:%.2f, %.2f, %.2f" % tuple(mAP3d[j, :3]), file=out)
        if compute_aos:
            print("aos  AP:%.2f, %.2f, %.2f" % tuple(mAPaos[j, :3]), file=out)
    return out.getvalue()


def filter_det_range(dets, close, far):
    """eval.py:897-913: the boxes of one frame whose |z_camera| lies in (close, far]."""
    from copy import deepcopy
    dets = deepcopy(dets)
    if dets['location'].shape[0] == 0:
        return dets
    valid_idx = (np.abs(dets['location'][:, 2]) > close) * (np.abs(dets['location'][:, 2]) <= far)
    for k in dets:
        if k == 'frame_id' or k == 'gt_boxes_lidar':
            continue
        dets[k] = dets[k][valid_idx]
    return dets


def get_range_eval_result(gt_annos, dt_annos, current_classes, PR_detail_dict=None, ranges=(0, 30, 50, 80)):
    """eval.py:916-1008 -> (report text, {'<Class>_<3d|bev>_iou<0.7|0.5>/<near>-<far>_R40': AP}): BEV / 3-D AP_R40 at
    difficulty 3 per distance band (and over all bands).  Class table and min-overlaps are the function's own, as in
    the reference (seven KITTI-style names by number)."""
    overlap_0_7 = np.array([[0.7, 0.5, 0.5, 0.7, 0.5, 0.7, 0.5]] * 3)
    overlap_0_5 = np.array([[0.7, 0.5, 0.5, 0.7, 0.5, 0.5, 0.25], [0.5, 0.25, 0.25, 0.5, 0.25, 0.5, 0.25],
                            [0.5, 0.25, 0.25, 0.5, 0.25, 0.5, 0.25]])
    min_overlaps = np.stack([overlap_0_7, overlap_0_5], axis=0)
    class_to_name = {0: 'Car', 1: 'Pedestrian', 2: 'Cyclist', 3: 'Van', 4: 'Person_sitting', 5: 'Truck', 6: 'Dynamic'}
    name_to_class = {v: n for n, v in class_to_name.items()}
    if not isinstance(current_classes, (list, tuple)):
        current_classes = [current_classes]
    current_classes = [name_to_class[c] if isinstance(c, str) else c for c in current_classes]
    min_overlaps = min_overlaps[:, :, current_classes]
    compute_aos = _compute_aos(dt_annos)
    ret_dict = {}
    range_pairs = [(ranges[i], ranges[i + 1]) for i in range(len(ranges) - 1)] + [(ranges[0], ranges[-1])]
    for range_s, range_e in range_pairs:
        dt_r = [filter_det_range(d, range_s, range_e) for d in dt_annos]
        gt_r = [filter_det_range(d, range_s, range_e) for d in gt_annos]
        res = do_eval(gt_r, dt_r, current_classes, min_overlaps, compute_aos, difficultys=[3], PR_detail_dict=PR_detail_dict)
        mAPbev_R40, mAP3d_R40 = res[5], res[6]
        for j, curcls in enumerate(current_classes):
            name = class_to_name[curcls]
            ret_dict['%s_3d_iou0.7/%02d-%02d_R40' % (name, range_s, range_e)] = mAP3d_R40[j, 0, 0]
            ret_dict['%s_3d_iou0.5/%02d-%02d_R40' % (name, range_s, range_e)] = mAP3d_R40[j, 0, 1]
            ret_dict['%s_bev_iou0.7/%02d-%02d_R40' % (name, range_s, range_e)] = mAPbev_R40[j, 0, 0]
            ret_dict['%s_bev_iou0.5/%02d-%02d_R40' % (name, range_s, range_e)] = mAPbev_R40[j, 0, 1]
    out = io.StringIO()
    head = "RANGE " + "  ".join("%02d-%02d " % rp for rp in range_pairs)
    for j, curcls in enumerate(current_classes):
        name = class_to_name[curcls]
        col = lambda kind, iou: [ret_dict['%s_%s_iou%s/%02d-%02d_R40' % (name, kind, iou, rs, re)] for rs, re in range_pairs]
        bev07, d07, bev05, d05 = col('bev', '0.7'), col('3d', '0.7'), col('bev', '0.5'), col('3d', '0.5')
        print("%s IoU 0.5:" % name, file=out)                        # (the reference's labels, eval.py:989-1006)
        print(head, file=out)
        print("BEV:  " + ", ".join("%6.3f" % x for x in bev07), file=out)
        print("3D :  " + ", ".join("%6.3f" % x for x in d07), file=out)
        print("%s IoU 0.25:" % name, file=out)
        print(head, file=out)
        print("BEV:  " + ", ".join("%6.3f" % x for x in bev05), file=out)
        print("3D :  " + ", ".join("%6.3f" % x for x in d05), file=out)
        print("%s IoU 0.7:" % name, file=out)
        print(", ".join("%3.1f / %3.1f" % (x, y) for x, y in zip(bev07, d07)), file=out)
        print("%s IoU 0.5:" % name, file=out)
        print(", ".join("%3.1f / %3.1f" % (x, y) for x, y in zip(bev05, d05)) + "\n", file=out)
    return out.getvalue(), ret_dict
