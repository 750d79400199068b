"""Drop-in for pcdet/ops/iou3d_nms/iou3d_nms_utils.py on the MI355X: the same function names, argument meaning and
return types (boxes are (.,7) [x, y, z, dx, dy, dz, heading] CUDA tensors), computed by libdfu3d_hip.so
(csrc/iou_stage.hip: exact polygon overlap, one wave per suppression-mask word)."""
import torch

from .. import stages as st


def _f32(boxes):
    if boxes.dim() != 2 or boxes.shape[1] != 7:
        raise ValueError("boxes must be (N, 7): x, y, z, dx, dy, dz, heading")
    return boxes.float().contiguous()


def boxes_iou_bev(boxes_a, boxes_b):
    """(N,7), (M,7) -> (N,M) IoU of the footprints in the ground plane   (iou3d_nms_utils.py:31-45)."""
    return st.boxes_bev(_f32(boxes_a), _f32(boxes_b), iou=True)


def _z_range(b):
    half = 0.5 * b[:, 5]
    return b[:, 2] - half, b[:, 2] + half


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """(N,7), (M,7) -> (N,M) volume IoU: footprint overlap x common z extent over the union   (:48-81)."""
    a, b = _f32(boxes_a), _f32(boxes_b)
    (a_lo, a_hi), (b_lo, b_hi) = _z_range(a), _z_range(b)
    common_z = (torch.minimum(a_hi[:, None], b_hi[None, :]) - torch.maximum(a_lo[:, None], b_lo[None, :])).clamp_(min=0)
    inter = st.boxes_bev(a, b, iou=False) * common_z
    volume = lambda t: t[:, 3] * t[:, 4] * t[:, 5]
    union = volume(a)[:, None] + volume(b)[None, :] - inter
    return inter / union.clamp_(min=1e-6)


def boxes_aligned_iou3d_gpu(boxes_a, boxes_b):
    """(N,7), (N,7) -> (N,1) volume IoU of row i with row i   (:84-117)."""
    a, b = _f32(boxes_a), _f32(boxes_b)
    if a.shape[0] != b.shape[0]:
        raise ValueError("aligned IoU needs as many boxes on both sides")
    (a_lo, a_hi), (b_lo, b_hi) = _z_range(a), _z_range(b)
    common_z = (torch.minimum(a_hi, b_hi) - torch.maximum(a_lo, b_lo)).clamp_(min=0)
    inter = st.boxes_bev_paired(a, b, iou=False) * common_z
    union = a[:, 3] * a[:, 4] * a[:, 5] + b[:, 3] * b[:, 4] * b[:, 5] - inter
    return (inter / union.clamp_(min=1e-6)).view(-1, 1)


def _nms(boxes, scores, thresh, pre_maxsize, normal):
    by_score = torch.argsort(scores, descending=True)
    if pre_maxsize is not None:
        by_score = by_score[:pre_maxsize]
    kept, n_kept = st.nms_bev(_f32(boxes[by_score]), float(thresh), normal=normal)
    return by_score[kept[:n_kept]].contiguous(), None


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """Rotated NMS -> (indices into `boxes` of the survivors, best score first; None)   (:120-135)."""
    return _nms(boxes, scores, thresh, pre_maxsize, normal=False)


def nms_normal_gpu(boxes, scores, thresh, **kwargs):
    """NMS on the axis-aligned footprints (headings ignored)   (:138-152)."""
    return _nms(boxes, scores, thresh, None, normal=True)
