"""The box conversions between a detector's LiDAR boxes and KITTI result rows (pcdet/utils/box_utils.py:203-288), used
by generate_prediction_dicts.  Host NumPy on a few hundred boxes per frame, as in the reference; same dtypes (float32
corner offsets, the rotation in the dtype of the boxes)."""
import numpy as np


def boxes3d_lidar_to_kitti_camera(boxes3d_lidar, calib):
    """(N,7) [x y z dx dy dz heading] (centre) -> (N,7) [x y z l h w ry] in the rectified camera frame, y at the box
    bottom (box_utils.py:203-219)."""
    b = np.array(boxes3d_lidar, copy=True)
    xyz = b[:, 0:3]
    l, w, h = b[:, 3:4], b[:, 4:5], b[:, 5:6]
    xyz[:, 2] -= h.reshape(-1) / 2
    xyz_cam = calib.lidar_to_rect(xyz)
    ry = -b[:, 6:7] - np.pi / 2
    return np.concatenate([xyz_cam, l, h, w, ry], axis=-1)


def boxes3d_to_corners3d_kitti_camera(boxes3d, bottom_center=True):
    """(N,7) camera boxes -> (N,8,3) corners, 0-3 the bottom face, 4-7 the top face (box_utils.py:222-265)."""
    n = boxes3d.shape[0]
    l, h, w = boxes3d[:, 3], boxes3d[:, 4], boxes3d[:, 5]
    sx = np.array([1, 1, -1, -1, 1, 1, -1, -1], np.float64) / 2.
    sz = np.array([1, -1, -1, 1, 1, -1, -1, 1], np.float64) / 2.
    x_c = (l[:, None] * sx[None, :]).astype(np.float32)
    z_c = (w[:, None] * sz[None, :]).astype(np.float32)
    if bottom_center:
        y_c = np.zeros((n, 8), np.float32)
        y_c[:, 4:8] = -h.reshape(n, 1).repeat(4, axis=1)
    else:
        y_c = (h[:, None] * np.array([1, 1, 1, 1, -1, -1, -1, -1], np.float64)[None, :] / 2.).astype(np.float32)
    ry = boxes3d[:, 6]
    zeros, ones = np.zeros(ry.size, np.float32), np.ones(ry.size, np.float32)
    R = np.transpose(np.array([[np.cos(ry), zeros, -np.sin(ry)], [zeros, ones, zeros], [np.sin(ry), zeros, np.cos(ry)]]),
                     (2, 0, 1))
    rotated = np.matmul(np.stack([x_c, y_c, z_c], axis=2), R)
    centre = boxes3d[:, 0:3].reshape(-1, 1, 3)
    return np.stack([centre[:, :, 0] + rotated[:, :, 0], centre[:, :, 1] + rotated[:, :, 1],
                     centre[:, :, 2] + rotated[:, :, 2]], axis=2).astype(np.float32)


def boxes3d_kitti_camera_to_imageboxes(boxes3d, calib, image_shape=None):
    """(N,7) camera boxes -> (N,4) [x1 y1 x2 y2]: the bounding rectangle of the eight projected corners, clipped to the
    image when its (height, width) is given (box_utils.py:268-288)."""
    corners = boxes3d_to_corners3d_kitti_camera(boxes3d)
    pts_img, _ = calib.rect_to_img(corners.reshape(-1, 3))
    uv = pts_img.reshape(-1, 8, 2)
    boxes = np.concatenate([np.min(uv, axis=1), np.max(uv, axis=1)], axis=1)
    if image_shape is not None:
        boxes[:, 0::2] = np.clip(boxes[:, 0::2], a_min=0, a_max=image_shape[1] - 1)
        boxes[:, 1::2] = np.clip(boxes[:, 1::2], a_min=0, a_max=image_shape[0] - 1)
    return boxes
