"""Mirror of the hot-path wrapper tools/PENet/vis_utils.py:136-166
(`save_depth_as_points`): reads one KITTI-format frame, FOV-filters the sweep,
obtains the instance masks, runs depth2pointsrgbpm and saves the float16
virtual-point file.  SEEM / X-Decoder (vis_utils.py:169-223) is an input
provider outside this path: masks come from `seg_provider` (default: a
precomputed `<root>/seg_2/<idx>.npz`)."""
import os

import numpy as np
import torch

from .. import stages as st
from ..calibration import Calibration
from ..params import Params
from .my_loader import depth2pointsrgbpm


def load_seg_npz(root_path, file_idx, seg_dir=None):
    """-> (thing_classes, masks (M,H,W), classes (M,), scores (M,), boxes2D (M,4))."""
    z = np.load(os.path.join(seg_dir or os.path.join(root_path, 'seg_2'), file_idx + '.npz'), allow_pickle=False)
    names = [str(s) for s in z['thing_classes']]
    keep = z['scores'] > 0.7                                          # vis_utils.py:218
    return names, z['masks'][keep], z['classes'][keep], z['scores'][keep], z['boxes'][keep]


def get_fov_flag_gpu(lidar, calib, img_shape, device="cuda:0"):
    """vis_utils.py:108-123 on the GPU (dfu3d_fov_filter) -> bool mask (n,)."""
    n = lidar.shape[0]
    dev = torch.device(device)
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a)).to(dt).to(dev).contiguous()
    idx = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    pts = t(lidar if n else np.zeros((1, 4), np.float32), torch.float32)
    st.fov_filter(pts, t([0, n], torch.int32), t([0], torch.int32), t(calib.record()[None], torch.float32),
                  1, (int(img_shape[0]), int(img_shape[1])), max(n, 1), idx, cnt, np.array([0, n]), np.array([0]))
    flag = np.zeros((n,), bool)
    flag[idx[:int(cnt.item())].cpu().numpy()] = True
    return flag


def save_depth_as_points(depth, idx, root_path, seg_provider=None, label_root=None, params=None,
                         crop_hw=None, device="cuda:0", return_rows=False, save_points=True, points_dir=None):
    """vis_utils.py:136-166.  `idx` is the frame id (int -> zero-padded to 6, or a
    file stem); `crop_hw` reproduces the shipped [:352,:1216] crop (hazard H11),
    default: no crop (canonical nuScenes 900x1600).  save_points=False: labels only -- the image is then used for
    its size alone, and a data set without image_2/ is labelled on the depth map's size; points_dir: where the
    float16 virtual-point file goes (default <root>/velodyne_depth, vis_utils.py:156)."""
    from PIL import Image
    p = params or Params()
    file_idx = str(idx).zfill(6)                                      # :137
    file_image_path = os.path.join(root_path, 'image_2', file_idx + '.png')
    file_velo_path = os.path.join(root_path, 'velodyne', file_idx + '.bin')
    file_calib = os.path.join(root_path, 'calib', file_idx + '.txt')
    calib = Calibration(file_calib)                                   # :142
    lidar = np.fromfile(str(file_velo_path), dtype=np.float32).reshape(-1, 4)   # :144
    if save_points or os.path.exists(file_image_path):
        image1 = np.array(Image.open(file_image_path).convert('RGB'), dtype=np.int32)   # :146
    else:                                # labels only, no image on disk: its size is the depth map's, its colours are not needed
        dshape = tuple(depth.shape)
        image1 = np.zeros((int(dshape[-2]), int(dshape[-1]), 3), np.int32)
    image = image1 if crop_hw is None else image1[:crop_hw[0], :crop_hw[1]]
    seg = (seg_provider or (lambda path: load_seg_npz(root_path, file_idx)))(file_image_path)
    thing_classes, masks, classes, scores, boxes2D = seg              # :150
    H, W = image.shape[0], image.shape[1]
    lidar = lidar[get_fov_flag_gpu(lidar, calib, (H, W), device)]     # :152-154: FOV test on the (cropped) image size
    paths = points_dir or os.path.join(root_path, 'velodyne_depth')   # :156-160
    if save_points:
        os.makedirs(paths, exist_ok=True)
    out_path = os.path.join(paths, file_idx + '.npy')
    if isinstance(depth, torch.Tensor):
        depth = depth.detach().cpu().numpy()
    depth = np.ascontiguousarray(depth, np.float32).reshape(-1)[:H * W].reshape(H, W, 1)   # :161
    if crop_hw is None:
        # canonical nuScenes (hazard H11): every size is the image size
        if tuple(p.bounds_hw) != (H, W) or tuple(p.fov_hw) != (H, W):
            p = Params(**{**p.__dict__, "bounds_hw": (H, W), "fov_hw": (H, W)})
    else:
        # the shipped mix: FOV filter and depth on the crop, masks at their own size, bounds as in `params`
        # (my_loader.py:526 hard-codes 1600x900)
        p = Params(**{**p.__dict__, "fov_hw": (H, W)})
    final_points, rows = depth2pointsrgbpm(depth, image, image1, calib, lidar, thing_classes,
                                           np.asarray(masks), classes, scores, boxes2D, None,
                                           file_idx, label_root=label_root or os.path.join(root_path, 'label_2'),
                                           params=p, device=device, return_rows=True)       # :163
    if save_points:
        np.save(out_path, final_points.astype(np.float16))            # :164-166
    return (out_path, rows) if return_rows else out_path
