"""Seeded annotation sets for the KITTI AP evaluator (row f-3): ground truth and detections in the dictionaries
kitti_common.get_label_annos / generate_prediction_dicts produce (name, truncated, occluded, alpha, bbox, dimensions
[l, h, w], location, rotation_y, score).  Shared by capture_eval_golden.py and the tests."""
import numpy as np

NAMES = ['Car', 'Truck', 'Construction_vehicle', 'Bus', 'Trailer', 'Barrier', 'Motorcycle', 'Bicycle', 'Pedestrian',
         'Traffic_cone']
# golden G9 (capture_eval_golden.py): name -> (seed, frames, keyword arguments of make_annos, classes evaluated)
GOLDEN_CASES = {
    'a': (11, 23, dict(max_gt=9, extra_dt=5, classes=('Car', 'Pedestrian', 'Truck')), ['Car', 'Pedestrian', 'Truck']),
    'b': (12, 9, dict(max_gt=6, extra_dt=3, classes=('Car', 'Bus'), with_alpha=False), ['Car', 'Bus']),
    'c': (13, 105, dict(max_gt=4, extra_dt=2, classes=('Car', 'Bicycle')), ['Car', 'Bicycle']),   # more frames than num_parts
}
SIZES = {'Car': (4.2, 1.6, 1.9), 'Truck': (7.0, 2.9, 2.5), 'Bus': (11.0, 3.4, 2.9), 'Pedestrian': (0.7, 1.8, 0.7),
         'Bicycle': (1.8, 1.3, 0.6), 'Van': (5.0, 2.0, 2.0), 'Person_sitting': (0.8, 1.2, 0.8)}


def _anno(n):
    return {'name': np.empty(n, dtype='<U24'), 'truncated': np.zeros(n), 'occluded': np.zeros(n, np.int64),
            'alpha': np.zeros(n), 'bbox': np.zeros((n, 4)), 'dimensions': np.zeros((n, 3)),
            'location': np.zeros((n, 3)), 'rotation_y': np.zeros(n), 'score': np.zeros(n)}


def make_annos(seed, frames, max_gt=9, extra_dt=5, classes=('Car', 'Pedestrian', 'Truck'), with_alpha=True,
               empty_frames=True):
    """-> (gt_annos, dt_annos).  Detections are jittered copies of most ground truths (some of another class, some
    dropped) plus unrelated boxes; 2-D boxes straddle the evaluator's height limits (25 / 40 px); Van / Person_sitting /
    DontCare rows exercise the ignore rules; a few frames have no ground truth or no detections."""
    rng = np.random.default_rng(seed)
    pool = list(classes) + ['Van', 'Person_sitting', 'DontCare']
    gts, dts = [], []
    for f in range(frames):
        n = int(rng.integers(0, max_gt + 1))
        if empty_frames and f % 7 == 3:
            n = 0
        g = _anno(n)
        for i in range(n):
            name = pool[int(rng.integers(0, len(pool)))] if rng.random() < 0.35 else classes[int(rng.integers(0, len(classes)))]
            g['name'][i] = name
            l, h, w = SIZES.get(name, (2.0, 1.5, 1.5))
            g['dimensions'][i] = np.array([l, h, w]) * rng.uniform(0.85, 1.15, 3)
            g['location'][i] = [rng.uniform(-30, 30), rng.uniform(1.0, 2.2), rng.uniform(4, 60)]
            g['rotation_y'][i] = rng.uniform(-np.pi, np.pi)
            g['alpha'][i] = rng.uniform(-np.pi, np.pi)
            x1, y1 = rng.uniform(0, 1400), rng.uniform(0, 700)
            hh = rng.choice([12.0, 24.0, 25.0, 26.0, 39.0, 40.0, 41.0, 80.0, 150.0]) + rng.choice([0.0, 0.25])
            g['bbox'][i] = [x1, y1, x1 + rng.uniform(15, 220), y1 + hh]
            g['truncated'][i] = rng.choice([0.0, 0.1, 0.15, 0.2, 0.3, 0.4, 0.5, 0.7])
            g['occluded'][i] = int(rng.integers(0, 4))
        rows = []
        for i in range(n):
            if g['name'][i] == 'DontCare' or rng.random() < 0.2:
                continue
            for _ in range(1 + int(rng.random() < 0.25)):                 # now and then two detections of one object
                name = g['name'][i] if rng.random() < 0.85 else classes[int(rng.integers(0, len(classes)))]
                if name in ('Van', 'Person_sitting'):
                    name = classes[0]
                jit = rng.choice([0.02, 0.1, 0.3, 0.6])
                rows.append((name, g['dimensions'][i] * rng.uniform(1 - 0.3 * jit, 1 + 0.3 * jit, 3),
                             g['location'][i] + rng.normal(0, jit, 3) * [1.0, 0.3, 1.0],
                             g['rotation_y'][i] + rng.normal(0, 0.3 * jit),
                             g['bbox'][i] + rng.normal(0, 12 * jit, 4), g['alpha'][i] + rng.normal(0, 0.2)))
        for _ in range(int(rng.integers(0, extra_dt + 1))):
            name = classes[int(rng.integers(0, len(classes)))]
            l, h, w = SIZES.get(name, (2.0, 1.5, 1.5))
            x1, y1 = rng.uniform(0, 1400), rng.uniform(0, 700)
            rows.append((name, np.array([l, h, w]), np.array([rng.uniform(-30, 30), rng.uniform(1.0, 2.2), rng.uniform(4, 60)]),
                         rng.uniform(-np.pi, np.pi), np.array([x1, y1, x1 + rng.uniform(15, 220), y1 + rng.uniform(10, 120)]),
                         rng.uniform(-np.pi, np.pi)))
        if empty_frames and f % 7 == 5:
            rows = []
        d = _anno(len(rows))
        for j, (name, dims, loc, ry, bbox, alpha) in enumerate(rows):
            d['name'][j] = name
            d['dimensions'][j] = dims
            d['location'][j] = loc
            d['rotation_y'][j] = ry
            d['bbox'][j] = bbox
            d['alpha'][j] = alpha if with_alpha else -10.0
            d['score'][j] = np.round(rng.uniform(0.05, 1.0), 3 if rng.random() < 0.5 else 6)   # equal scores happen
        gts.append(g)
        dts.append(d)
    return gts, dts


def pack(annos):
    """annotation list -> flat arrays for an .npz (names as an array of strings, per-frame counts)."""
    keys = ('truncated', 'occluded', 'alpha', 'bbox', 'dimensions', 'location', 'rotation_y', 'score')
    out = {'count': np.array([len(a['name']) for a in annos], np.int64),
           'name': np.concatenate([np.asarray(a['name'], dtype='<U24') for a in annos]) if annos else np.zeros(0, '<U24')}
    for k in keys:
        out[k] = np.concatenate([np.asarray(a[k]) for a in annos], 0)
    return out


def unpack(flat):
    annos, at = [], 0
    for n in flat['count']:
        n = int(n)
        annos.append({k: np.asarray(flat[k][at:at + n]) for k in flat if k != 'count'})
        at += n
    return annos
