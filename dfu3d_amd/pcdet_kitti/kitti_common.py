"""Drop-in for the label readers of pcdet/datasets/kitti/kitti_object_eval_python/kitti_common.py: `get_label_anno`
(:294-330) and `get_label_annos` (:332-347) -- KITTI label / result files -> the annotation dictionaries the evaluator
(eval.py) works on.  Host code, as in the reference."""
import pathlib
import re

import numpy as np


def get_label_anno(label_path):
    """One label or result file -> {'name', 'truncated', 'occluded', 'alpha', 'bbox' (n,4), 'dimensions' (n,3) as
    [l, h, w] (the file holds h w l), 'location' (n,3), 'rotation_y', 'score' (zeros when the file has 15 columns)}."""
    with open(label_path, 'r') as f:
        rows = [line.strip().split(' ') for line in f.readlines()]
    col = lambda a, b: np.array([[float(v) for v in r[a:b]] for r in rows], np.float64).reshape(-1, b - a)
    anno = {
        'name': np.array([r[0] for r in rows]),
        'truncated': np.array([float(r[1]) for r in rows]),
        'occluded': np.array([int(r[2]) for r in rows]),
        'alpha': np.array([float(r[3]) for r in rows]),
        'bbox': col(4, 8),
        'dimensions': col(8, 11)[:, [2, 0, 1]],
        'location': col(11, 14),
        'rotation_y': np.array([float(r[14]) for r in rows]).reshape(-1),
    }
    if rows and len(rows[0]) == 16:
        anno['score'] = np.array([float(r[15]) for r in rows])
    else:
        anno['score'] = np.zeros([anno['bbox'].shape[0]])
    return anno


def get_label_annos(label_folder, image_ids=None):
    """All `NNNNNN.txt` files of a folder (or the given ids; an int n means 0..n-1), in ascending id order."""
    folder = pathlib.Path(label_folder)
    if image_ids is None:
        image_ids = sorted(int(p.stem) for p in folder.glob('*.txt') if re.match(r'^\d{6}.txt$', p.name))
    if not isinstance(image_ids, list):
        image_ids = list(range(image_ids))
    return [get_label_anno(folder / ('%06d.txt' % int(i))) for i in image_ids]


def filter_annos_low_score(image_annos, thresh):
    """kitti_common.py:191-201: drop detections whose score is below `thresh`."""
    out = []
    for anno in image_annos:
        keep = np.asarray(anno['score']) >= thresh
        out.append({k: np.asarray(v)[keep] for k, v in anno.items()})
    return out
