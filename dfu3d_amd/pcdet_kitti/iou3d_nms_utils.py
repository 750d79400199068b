"""Mirror of pcdet/ops/iou3d_nms/iou3d_nms_utils.py (boxes_iou_bev, boxes_iou3d_gpu, nms_gpu) on the MI355X."""
import torch

from .. import stages as st


def boxes_iou_bev(boxes_a, boxes_b):
    """(N,7),(M,7) CUDA float tensors -> (N,M) BEV IoU   (iou3d_nms_utils.py:31-45)."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    return st.boxes_bev(boxes_a.float().contiguous(), boxes_b.float().contiguous(), iou=True)


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """(N,7),(M,7) -> (N,M) 3-D IoU   (iou3d_nms_utils.py:48-81, same torch expressions around the overlap kernel)."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    boxes_a, boxes_b = boxes_a.float(), boxes_b.float()
    boxes_a_height_max = (boxes_a[:, 2] + boxes_a[:, 5] / 2).view(-1, 1)
    boxes_a_height_min = (boxes_a[:, 2] - boxes_a[:, 5] / 2).view(-1, 1)
    boxes_b_height_max = (boxes_b[:, 2] + boxes_b[:, 5] / 2).view(1, -1)
    boxes_b_height_min = (boxes_b[:, 2] - boxes_b[:, 5] / 2).view(1, -1)
    overlaps_bev = st.boxes_bev(boxes_a.contiguous(), boxes_b.contiguous(), iou=False)
    max_of_min = torch.max(boxes_a_height_min, boxes_b_height_min)
    min_of_max = torch.min(boxes_a_height_max, boxes_b_height_max)
    overlaps_h = torch.clamp(min_of_max - max_of_min, min=0)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(1, -1)
    return overlaps_3d / torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-6)


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """
    :param boxes: (N, 7) [x, y, z, dx, dy, dz, heading]
    :param scores: (N)
    :param thresh:
    :return: (indices of the kept boxes in `boxes`, best score first; None)   (iou3d_nms_utils.py:120-135)
    """
    assert boxes.shape[1] == 7
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    boxes = boxes[order].float().contiguous()
    keep, num_out = st.nms_bev(boxes, float(thresh))
    return order[keep[:num_out]].contiguous(), None
