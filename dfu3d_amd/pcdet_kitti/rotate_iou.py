"""Drop-in for pcdet/datasets/kitti/kitti_object_eval_python/rotate_iou.py: `rotate_iou_gpu_eval`, the rotated-box IoU
of the KITTI AP evaluator (eval.py:122,156).  The reference JIT-compiles a CUDA kernel through numba.cuda, which has no
ROCm target -- this is what blocks the AP step of the self-evolution loop on an MI355X.  Same signature and return type
(NumPy in, NumPy float32 out); the overlap is the exact intersection polygon (csrc/iou_stage.hip)."""
import numpy as np
import torch

from .. import stages as st


def rotate_iou_gpu_eval(boxes, query_boxes, criterion=-1, device_id=0):
    """boxes (N,5), query_boxes (K,5): [centre x, centre y, width, height, angle] -> (N,K) float32.
    criterion -1: IoU; 0: overlap / area(box); 1: overlap / area(query box); anything else: the overlap area
    (rotate_iou.py:247-255, 293-330)."""
    boxes = np.ascontiguousarray(boxes, np.float32).reshape(-1, 5)
    query_boxes = np.ascontiguousarray(query_boxes, np.float32).reshape(-1, 5)
    n, k = boxes.shape[0], query_boxes.shape[0]
    if n == 0 or k == 0:
        return np.zeros((n, k), np.float32)
    dev = torch.device("cuda", device_id)
    out = st.rotate_iou_eval(torch.from_numpy(boxes).to(dev), torch.from_numpy(query_boxes).to(dev), criterion)
    return out.cpu().numpy()
