"""Drop-in for pcdet/datasets/kitti/kitti_object_eval_python/evaluate.py: result folder + label folder + split file ->
the official AP report, on the GPU.

    python -m dfu3d_amd.pcdet_kitti.evaluate --label_path <label_2> --result_path <results> \\
        --label_split_file <val.txt> --current_class Car Pedestrian [--score_thresh 0.1]

(the reference exposes the same `evaluate` function through `fire`; its `coco=True` variant is not built)."""
import argparse

from . import eval as kitti_eval
from . import kitti_common as kitti


def _read_imageset_file(path):
    with open(path, 'r') as f:
        return [int(line) for line in f.readlines() if line.strip()]


def evaluate(label_path, result_path, label_split_file, current_class=0, coco=False, score_thresh=-1):
    """evaluate.py:15-30 -> (report text, AP dictionary).  The result files are read for the ids of the split file (the
    reference reads every file of the result folder and relies on the folder holding exactly the split)."""
    if coco:
        raise NotImplementedError("get_coco_eval_result is not built (unused by KittiDataset.evaluation)")
    val_image_ids = _read_imageset_file(label_split_file)
    dt_annos = kitti.get_label_annos(result_path, val_image_ids)
    if score_thresh > 0:
        dt_annos = kitti.filter_annos_low_score(dt_annos, score_thresh)
    gt_annos = kitti.get_label_annos(label_path, val_image_ids)
    return kitti_eval.get_official_eval_result(gt_annos, dt_annos, current_class)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('--label_path', required=True)
    ap.add_argument('--result_path', required=True)
    ap.add_argument('--label_split_file', required=True)
    ap.add_argument('--current_class', nargs='+', default=['Car'])
    ap.add_argument('--score_thresh', type=float, default=-1)
    a = ap.parse_args(argv)
    classes = [int(c) if c.isdigit() else c for c in a.current_class]
    text, _ = evaluate(a.label_path, a.result_path, a.label_split_file, classes, score_thresh=a.score_thresh)
    print(text)
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
