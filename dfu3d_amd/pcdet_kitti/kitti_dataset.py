"""The two ends of the self-evolution loop's evaluation step, mirroring the static / instance
methods of pcdet/datasets/kitti/kitti_dataset.py as functions:
  * generate_prediction_dicts (:339-419): detector output (LiDAR boxes, scores, labels) -> KITTI annotation dictionaries
    and, optionally, result files `<frame_id>.txt`;
  * evaluation (:421-431): annotation dictionaries -> the official AP report, computed on the GPU (eval.py of this
    package) where the reference needs numba-CUDA.
"""
import copy
import os

import numpy as np

from . import box_utils
from . import eval as kitti_eval

_ROW = '%s -1 -1 %.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f'


def _to_numpy(t):
    return t.detach().cpu().numpy() if hasattr(t, 'detach') else np.asarray(t)


def _empty_prediction(n):
    return {'name': np.zeros(n), 'truncated': np.zeros(n), 'occluded': np.zeros(n), 'alpha': np.zeros(n),
            'bbox': np.zeros([n, 4]), 'dimensions': np.zeros([n, 3]), 'location': np.zeros([n, 3]),
            'rotation_y': np.zeros(n), 'score': np.zeros(n), 'boxes_lidar': np.zeros([n, 7])}


def generate_prediction_dicts(batch_dict, pred_dicts, class_names, output_path=None):
    """batch_dict: 'frame_id', 'calib', 'image_shape' per sample; pred_dicts: per sample 'pred_boxes' (N,7),
    'pred_scores' (N), 'pred_labels' (N, 1-based).  -> list of annotation dictionaries (+ 'frame_id', 'boxes_lidar')."""
    annos = []
    for index, box_dict in enumerate(pred_dicts):
        scores = _to_numpy(box_dict['pred_scores'])
        boxes = _to_numpy(box_dict['pred_boxes'])
        labels = _to_numpy(box_dict['pred_labels'])
        pred = _empty_prediction(scores.shape[0])
        if scores.shape[0] != 0:
            calib = batch_dict['calib'][index]
            image_shape = _to_numpy(batch_dict['image_shape'][index])
            cam = box_utils.boxes3d_lidar_to_kitti_camera(boxes, calib)
            pred['name'] = np.array(class_names)[labels - 1]
            pred['alpha'] = -np.arctan2(-boxes[:, 1], boxes[:, 0]) + cam[:, 6]
            pred['bbox'] = box_utils.boxes3d_kitti_camera_to_imageboxes(cam, calib, image_shape=image_shape)
            pred['dimensions'] = cam[:, 3:6]
            pred['location'] = cam[:, 0:3]
            pred['rotation_y'] = cam[:, 6]
            pred['score'] = scores
            pred['boxes_lidar'] = boxes
        pred['frame_id'] = batch_dict['frame_id'][index]
        annos.append(pred)
        if output_path is not None:
            with open(os.path.join(str(output_path), '%s.txt' % pred['frame_id']), 'w') as f:
                bbox, loc, dims = pred['bbox'], pred['location'], pred['dimensions']      # dims: l h w -> file: h w l
                for k in range(len(bbox)):
                    print(_ROW % (pred['name'][k], pred['alpha'][k], bbox[k][0], bbox[k][1], bbox[k][2], bbox[k][3],
                                  dims[k][1], dims[k][2], dims[k][0], loc[k][0], loc[k][1], loc[k][2],
                                  pred['rotation_y'][k], pred['score'][k]), file=f)
    return annos


def evaluation(det_annos, class_names, gt_annos):
    """kitti_dataset.py:421-431 with the ground-truth annotations handed in (the reference takes them from its info
    file) -> (report text, AP dictionary)."""
    if gt_annos is None:
        return None, {}
    return kitti_eval.get_official_eval_result(copy.deepcopy(list(gt_annos)), copy.deepcopy(list(det_annos)), class_names)
