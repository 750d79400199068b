"""Mirror of pcdet/ops/roiaware_pool3d/roiaware_pool3d_utils.py:9-25 on the GPU."""
import numpy as np
import torch

from .. import stages as st


def points_in_boxes_cpu(points, boxes, device="cuda:0"):
    """
    Args:
        points: (num_points, 3)
        boxes: [x, y, z, dx, dy, dz, heading], (x, y, z) is the box center
    Returns:
        point_indices: (N, num_points) int32, numpy in -> numpy out (same contract as the reference;
        the name is the reference's, the work runs in dfu3d_points_in_boxes_mask)
    """
    is_numpy = isinstance(points, np.ndarray)
    p = torch.as_tensor(points)
    b = torch.as_tensor(boxes)
    assert b.shape[1] == 7
    assert p.shape[1] == 3
    if p.shape[0] == 0 or b.shape[0] == 0:
        out = torch.zeros((b.shape[0], p.shape[0]), dtype=torch.int32)
        return out.numpy() if is_numpy else out
    out = st.points_in_boxes_mask(p.to(device=device, dtype=torch.float32).contiguous(),
                                  b.to(device=device, dtype=torch.float64).contiguous())
    return out.cpu().numpy() if is_numpy else out
