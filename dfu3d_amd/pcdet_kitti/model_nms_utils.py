"""Drop-in for pcdet/models/model_utils/model_nms_utils.py: the detector's post-processing on top of the rotated NMS
of this package (iou3d_nms_utils -> csrc/iou_stage.hip).  `nms_config` is the reference's EasyDict (NMS_TYPE,
NMS_THRESH, NMS_PRE_MAXSIZE, NMS_POST_MAXSIZE; cfg centerpoint_nuscenes2kitti.yaml:87-93) or any mapping / object with
those fields."""
import torch

from . import iou3d_nms_utils


def _cfg(nms_config, key):
    return nms_config[key] if isinstance(nms_config, dict) else getattr(nms_config, key)


def _select(scores, boxes, nms_config):
    """Indices (into scores / boxes) that survive: the NMS_PRE_MAXSIZE best by score, suppressed, cut to NMS_POST_MAXSIZE."""
    if scores.shape[0] == 0:
        return []
    top_scores, order = torch.topk(scores, k=min(_cfg(nms_config, 'NMS_PRE_MAXSIZE'), scores.shape[0]))
    extra = dict(nms_config) if isinstance(nms_config, dict) else dict(vars(nms_config))
    keep, _ = getattr(iou3d_nms_utils, _cfg(nms_config, 'NMS_TYPE'))(boxes[order][:, 0:7], top_scores,
                                                                   _cfg(nms_config, 'NMS_THRESH'), **extra)
    return order[keep[:_cfg(nms_config, 'NMS_POST_MAXSIZE')]]


def class_agnostic_nms(box_scores, box_preds, nms_config, score_thresh=None):
    """model_nms_utils.py:6-26 -> (indices into the input, their scores)."""
    all_scores = box_scores
    if score_thresh is not None:
        above = box_scores >= score_thresh
        box_scores, box_preds = box_scores[above], box_preds[above]
    selected = _select(box_scores, box_preds, nms_config)
    if score_thresh is not None:
        selected = above.nonzero().view(-1)[selected]
    return selected, all_scores[selected]


def multi_classes_nms(cls_scores, box_preds, nms_config, score_thresh=None):
    """model_nms_utils.py:29-66: NMS per class column -> (scores, 0-based labels, boxes), classes one after the other."""
    scores, labels, boxes = [], [], []
    for k in range(cls_scores.shape[1]):
        col, cur = cls_scores[:, k], box_preds
        if score_thresh is not None:
            above = col >= score_thresh
            col, cur = col[above], box_preds[above]
        selected = _select(col, cur, nms_config)
        scores.append(col[selected])
        labels.append(col.new_ones(len(selected)).long() * k)
        boxes.append(cur[selected])
    return torch.cat(scores, dim=0), torch.cat(labels, dim=0), torch.cat(boxes, dim=0)


def class_specific_nms(box_scores, box_preds, box_labels, nms_config, score_thresh=None):
    """model_nms_utils.py:69-105: per-class thresholds and sizes (lists in nms_config) -> (indices, scores)."""
    selected = []
    for k in range(len(_cfg(nms_config, 'NMS_THRESH'))):
        mask = box_labels == k
        if isinstance(score_thresh, float):
            mask = mask & (box_scores > score_thresh)
        elif isinstance(score_thresh, list):
            mask = mask & (box_scores > score_thresh[k])
        idx = torch.nonzero(mask)[:, 0]
        if idx.shape[0] > 0:
            keep, _ = iou3d_nms_utils.nms_gpu(box_preds[mask], box_scores[mask], thresh=_cfg(nms_config, 'NMS_THRESH')[k],
                                              pre_maxsize=_cfg(nms_config, 'NMS_PRE_MAXSIZE')[k],
                                              post_max_size=_cfg(nms_config, 'NMS_POST_MAXSIZE')[k])
            selected.append(idx[keep])
    if len(selected) != 0:
        selected = torch.cat(selected)
    return selected, box_scores[selected]
