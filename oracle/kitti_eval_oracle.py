"""CPU checker for SURVEY.md §8 row f-3, AP-evaluation half -- TEST INFRASTRUCTURE ONLY.

Only tests/ and the cpu_baseline leg of tools/bench_eval.py may import this module.

A plain-Python restatement of the KITTI AP evaluator the reference ships in
pcdet/datasets/kitti/kitti_object_eval_python/eval.py (numba-jitted CPU loops around a numba-CUDA IoU kernel):
`clean_data` (:30-92), `image_box_overlap` (:95-124), `d3_box_overlap(_kernel)` (:126-158), `compute_statistics_jit`
(:161-290), `get_thresholds` (:9-27), `eval_class` (:447-577), `get_mAP` / `get_mAP_R40` (:580-591), `do_eval` (:645-690),
`get_official_eval_result` (:708-845; every class at 0.5 / 0.5 / 0.25 as the reference edits the thresholds).
Every frame is walked one after the other; there is no partitioning (`num_parts` only bounds the size of the reference's
dense cross-frame matrices and does not change any result).

Pinned: tests/golden/g9_kitti_eval.npz holds the outputs of the reference's own eval.py (imported with `numba.jit` as the
identity decorator -- numba is absent from this image -- tests/golden/capture_eval_golden.py) on seeded annotation sets;
tests/test_oracle_kitti_eval.py requires this module to reproduce them.  The rotated overlap of metric 1 / 2
(`rotate_iou_gpu_eval`, numba-CUDA: cannot run here) is oracle/iou3d_oracle.py's exact float64 polygon area rounded to
float32 in both the capture and here -- that leaf stays unpinned, as DESIGN.md says for the whole of row f-3.
"""
import math

import numpy as np

from . import iou3d_oracle

CLASS_NAMES = ['Car', 'Truck', 'Construction_vehicle', 'Bus', 'Trailer', 'Barrier', 'Motorcycle', 'Bicycle',
               'Pedestrian', 'Traffic_cone']                                       # eval.py:32
MIN_HEIGHT = [40, 25, 25]
MAX_OCCLUSION = [0, 1, 2]
MAX_TRUNCATION = [0.15, 0.3, 0.5]
N_SAMPLE_PTS = 41


def get_thresholds(scores, num_gt, num_sample_pts=N_SAMPLE_PTS):
    """eval.py:9-27: the scores at which the recall passes the sample points."""
    scores = sorted((float(s) for s in scores), reverse=True)
    out, current = [], 0.0
    n = len(scores)
    for i, s in enumerate(scores):
        left = (i + 1) / num_gt
        right = (i + 2) / num_gt if i < n - 1 else left
        if (right - current) < (current - left) and i < n - 1:
            continue
        out.append(s)
        current += 1 / (num_sample_pts - 1.0)
    return out


def clean_data(gt, dt, current_class, difficulty):
    """eval.py:30-92 -> (valid gt count, ignored_gt, ignored_dt, DontCare boxes)."""
    want = CLASS_NAMES[current_class].lower()
    ign_gt, ign_dt, dc = [], [], []
    n_valid = 0
    for i in range(len(gt['name'])):
        name = str(gt['name'][i]).lower()
        box = gt['bbox'][i]
        if name == want:
            cls = 1
        elif (want == 'pedestrian' and name == 'person_sitting') or (want == 'car' and name == 'van'):
            cls = 0
        else:
            cls = -1
        hard = False
        if difficulty < 3:
            hard = bool(gt['occluded'][i] > MAX_OCCLUSION[difficulty] or gt['truncated'][i] > MAX_TRUNCATION[difficulty]
                        or (box[3] - box[1]) <= MIN_HEIGHT[difficulty])
        if cls == 1 and not hard:
            ign_gt.append(0)
            n_valid += 1
        elif cls == 0 or (hard and cls == 1):
            ign_gt.append(1)
        else:
            ign_gt.append(-1)
        if str(gt['name'][i]) == 'DontCare':
            dc.append(np.asarray(box, np.float64))
    for i in range(len(dt['name'])):
        cls = 1 if str(dt['name'][i]).lower() == want else -1
        h = abs(dt['bbox'][i][3] - dt['bbox'][i][1])
        if difficulty in (0, 1, 2) and h < MIN_HEIGHT[difficulty]:
            ign_dt.append(1)
        elif cls == 1:
            ign_dt.append(0)
        else:
            ign_dt.append(-1)
    return n_valid, ign_gt, ign_dt, (np.stack(dc, 0) if dc else np.zeros((0, 4)))


def image_box_overlap(boxes, query, criterion=-1):
    """eval.py:95-124, float64."""
    boxes = np.asarray(boxes, np.float64).reshape(-1, 4)
    query = np.asarray(query, np.float64).reshape(-1, 4)
    out = np.zeros((len(boxes), len(query)))
    for k in range(len(query)):
        qa = (query[k, 2] - query[k, 0]) * (query[k, 3] - query[k, 1])
        for n in range(len(boxes)):
            iw = min(boxes[n, 2], query[k, 2]) - max(boxes[n, 0], query[k, 0])
            if iw <= 0:
                continue
            ih = min(boxes[n, 3], query[k, 3]) - max(boxes[n, 1], query[k, 1])
            if ih <= 0:
                continue
            ba = (boxes[n, 2] - boxes[n, 0]) * (boxes[n, 3] - boxes[n, 1])
            ua = {-1: ba + qa - iw * ih, 0: ba, 1: qa}.get(criterion, 1.0)
            out[n, k] = iw * ih / ua
    return out


def _cam_boxes(a):
    return np.concatenate([np.asarray(a['location'], np.float64).reshape(-1, 3),
                           np.asarray(a['dimensions'], np.float64).reshape(-1, 3),
                           np.asarray(a['rotation_y'], np.float64).reshape(-1, 1)], 1)


def rotated_overlap_f32(boxes5, query5, criterion):
    """Stand-in for rotate_iou_gpu_eval (rotate_iou.py:293-330): exact polygon area, handed over as float32."""
    if len(boxes5) == 0 or len(query5) == 0:
        return np.zeros((len(boxes5), len(query5)), np.float32)
    return iou3d_oracle.rotate_iou_eval(np.asarray(boxes5, np.float32), np.asarray(query5, np.float32),
                                        criterion).astype(np.float32)


def frame_overlaps(gt, dt, metric):
    """(detections, ground truths) overlap matrix of one frame as eval_class sees it (eval.py:483: the evaluator hands
    the detections in as `gt_annos`), float64."""
    if metric == 0:
        return image_box_overlap(dt['bbox'], gt['bbox'])
    d, g = _cam_boxes(dt), _cam_boxes(gt)
    if metric == 1:                                                   # eval.py:367-381: x, z, l, w, ry
        return rotated_overlap_f32(d[:, [0, 2, 3, 5, 6]], g[:, [0, 2, 3, 5, 6]], -1).astype(np.float64)
    rinc = rotated_overlap_f32(d[:, [0, 2, 3, 5, 6]], g[:, [0, 2, 3, 5, 6]], 2)          # the overlap area (:154-158)
    for i in range(len(d)):                                           # eval.py:126-151 (y points down, y = box bottom)
        for j in range(len(g)):
            if rinc[i, j] > 0:
                iw = min(d[i, 1], g[j, 1]) - max(d[i, 1] - d[i, 4], g[j, 1] - g[j, 4])
                if iw > 0:
                    inc = iw * float(rinc[i, j])
                    ua = d[i, 3] * d[i, 4] * d[i, 5] + g[j, 3] * g[j, 4] * g[j, 5] - inc
                    rinc[i, j] = inc / ua                             # stored into the float32 matrix
                else:
                    rinc[i, j] = 0.0
    return rinc.astype(np.float64)


def compute_statistics(overlaps, gt_alpha, dt_bbox, dt_alpha, dt_score, ign_gt, ign_dt, dc, metric, min_overlap,
                       thresh=0.0, compute_fp=False, compute_aos=False):
    """eval.py:161-290 -> (tp, fp, fn, similarity, scores of the matched detections)."""
    D, G = len(ign_dt), len(ign_gt)
    assigned = [False] * D
    below = [bool(compute_fp and dt_score[j] < thresh) for j in range(D)]
    NONE = -10000000
    tp = fp = fn = 0
    similarity = 0
    matched, delta = [], []
    for i in range(G):
        if ign_gt[i] == -1:
            continue
        pick, valid, best, picked_ignored = -1, NONE, 0, False
        for j in range(D):
            if ign_dt[j] == -1 or assigned[j] or below[j]:
                continue
            ov = overlaps[j, i]
            if not compute_fp and ov > min_overlap and dt_score[j] > valid:
                pick, valid = j, dt_score[j]
            elif compute_fp and ov > min_overlap and (ov > best or picked_ignored) and ign_dt[j] == 0:
                best, pick, valid, picked_ignored = ov, j, 1, False
            elif compute_fp and ov > min_overlap and valid == NONE and ign_dt[j] == 1:
                pick, valid, picked_ignored = j, 1, True
        if valid == NONE and ign_gt[i] == 0:
            fn += 1
        elif valid != NONE and (ign_gt[i] == 1 or ign_dt[pick] == 1):
            assigned[pick] = True
        elif valid != NONE:
            tp += 1
            matched.append(float(dt_score[pick]))
            if compute_aos:
                delta.append(float(gt_alpha[i]) - float(dt_alpha[pick]))
            assigned[pick] = True
    if compute_fp:
        for j in range(D):
            if not (assigned[j] or ign_dt[j] == -1 or ign_dt[j] == 1 or below[j]):
                fp += 1
        stuff = 0
        if metric == 0:
            odc = image_box_overlap(dt_bbox, dc, 0)
            for i in range(len(dc)):
                for j in range(D):
                    if assigned[j] or ign_dt[j] == -1 or ign_dt[j] == 1 or below[j]:
                        continue
                    if odc[j, i] > min_overlap:
                        assigned[j] = True
                        stuff += 1
        fp -= stuff
        if compute_aos:
            if tp > 0 or fp > 0:
                similarity = float(np.sum(np.array([0.0] * fp + [(1.0 + math.cos(d)) / 2.0 for d in delta])))
            else:
                similarity = -1
    return tp, fp, fn, similarity, matched


def eval_class(gt_annos, dt_annos, current_classes, difficultys, metric, min_overlaps, compute_aos=False):
    """eval.py:447-577 -> {'recall', 'precision', 'orientation'}: (class, difficulty, min_overlap, 41)."""
    assert len(gt_annos) == len(dt_annos)
    F = len(gt_annos)
    overlaps = [frame_overlaps(gt_annos[f], dt_annos[f], metric) for f in range(F)]
    shape = [len(current_classes), len(difficultys), len(min_overlaps), N_SAMPLE_PTS]
    precision, recall, aos = np.zeros(shape), np.zeros(shape), np.zeros(shape)
    for m, cls in enumerate(current_classes):
        for l, diff in enumerate(difficultys):
            cleaned = [clean_data(gt_annos[f], dt_annos[f], cls, diff) for f in range(F)]
            n_valid = sum(c[0] for c in cleaned)
            for k, min_overlap in enumerate(min_overlaps[:, metric, m]):
                def stats(f, **kw):
                    g, d = gt_annos[f], dt_annos[f]
                    return compute_statistics(overlaps[f], np.asarray(g['alpha'], np.float64),
                                              np.asarray(d['bbox'], np.float64).reshape(-1, 4),
                                              np.asarray(d['alpha'], np.float64), np.asarray(d['score'], np.float64),
                                              cleaned[f][1], cleaned[f][2], cleaned[f][3], metric, min_overlap, **kw)
                scores = []
                for f in range(F):
                    scores += stats(f, thresh=0.0, compute_fp=False)[4]
                thresholds = get_thresholds(scores, n_valid)
                pr = np.zeros([len(thresholds), 4])
                for f in range(F):
                    for t, th in enumerate(thresholds):
                        tp, fp, fn, sim, _ = stats(f, thresh=th, compute_fp=True, compute_aos=compute_aos)
                        pr[t, 0] += tp
                        pr[t, 1] += fp
                        pr[t, 2] += fn
                        if sim != -1:
                            pr[t, 3] += sim
                with np.errstate(invalid='ignore', divide='ignore'):
                    for i in range(len(thresholds)):
                        recall[m, l, k, i] = pr[i, 0] / (pr[i, 0] + pr[i, 2])
                        precision[m, l, k, i] = pr[i, 0] / (pr[i, 0] + pr[i, 1])
                        if compute_aos:
                            aos[m, l, k, i] = pr[i, 3] / (pr[i, 0] + pr[i, 1])
                for i in range(len(thresholds)):
                    precision[m, l, k, i] = np.max(precision[m, l, k, i:], axis=-1)
                    recall[m, l, k, i] = np.max(recall[m, l, k, i:], axis=-1)
                    if compute_aos:
                        aos[m, l, k, i] = np.max(aos[m, l, k, i:], axis=-1)
    return {'recall': recall, 'precision': precision, 'orientation': aos}


def get_mAP(prec):
    return sum(prec[..., i] for i in range(0, prec.shape[-1], 4)) / 11 * 100


def get_mAP_R40(prec):
    return sum(prec[..., i] for i in range(1, prec.shape[-1])) / 40 * 100


def official_min_overlaps():
    """eval.py:719-733: every class 0.5 / 0.5 / 0.5, the same again, then 0.25 -- (3, metric, class)."""
    return np.stack([np.full((3, 10), 0.5), np.full((3, 10), 0.5), np.full((3, 10), 0.25)], 0)


def official_result(gt_annos, dt_annos, current_classes):
    """The numbers behind get_official_eval_result's ret_dict (eval.py:708-845): '<Class>_<3d|bev|image|aos>/<level>_R40'."""
    name_to_class = {n: i for i, n in enumerate(CLASS_NAMES)}
    classes = [name_to_class[c] if isinstance(c, str) else c for c in current_classes]
    mo = official_min_overlaps()[:, :, classes]
    compute_aos = False
    for a in dt_annos:
        if len(a['alpha']) != 0:
            compute_aos = a['alpha'][0] != -10
            break
    out = {}
    r = eval_class(gt_annos, dt_annos, classes, (0, 1, 2), 0, mo, compute_aos)
    tables = {'image': get_mAP_R40(r['precision'])}
    if compute_aos:
        tables['aos'] = get_mAP_R40(r['orientation'])
    tables['bev'] = get_mAP_R40(eval_class(gt_annos, dt_annos, classes, (0, 1, 2), 1, mo)['precision'])
    tables['3d'] = get_mAP_R40(eval_class(gt_annos, dt_annos, classes, (0, 1, 2), 2, mo)['precision'])
    for j, c in enumerate(classes):
        for key, tab in tables.items():
            for l, level in enumerate(('easy', 'moderate', 'hard')):
                out['%s_%s/%s_R40' % (CLASS_NAMES[c], key, level)] = tab[j, l, 0]
    return out
