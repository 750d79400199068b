"""Mirror of tools/PENet/dataloaders/my_loader.py: `depth2pointsrgbpm` with the
reference's positional arguments (my_loader.py:502) -- one camera view in, one
KITTI label file out, virtual points returned -- computed on the GPU through
the C ABI.  Differences, all additive: the label directory is an argument
(the reference hard-codes it, my_loader.py:503, hazard H14), `plane` may be
supplied (hazard H1), and there is no 15 s clustering abort (hazard H2)."""
import os

import numpy as np
import torch

from ..engine import PseudoBoxEngine, ViewBatch
from ..labels import write_label_file
from ..params import NUSC_CLASSES, Params

_ENGINES = {}


def _engine(params, H, W, M, cap_n, device):
    import dataclasses
    key = (dataclasses.astuple(params), H, W, M, str(device))
    e = _ENGINES.get(key)
    if e is None or e.cap_n < cap_n:
        if len(_ENGINES) >= 2:          # every engine owns a bin table of ~37 MB x the canvas share: keep few
            _ENGINES.clear()
        e = PseudoBoxEngine(params, H, W, M, max(cap_n, 1024), views_per_chunk=1, dense=True,
                            cap_vox=1 << 19, pool_per_view=1 << 20, rows_per_view=1024,
                            device=device, apply_fov=False)
        _ENGINES[key] = e
    return e


def _boxes_array(boxes2D, M):
    if isinstance(boxes2D, (np.ndarray, torch.Tensor)):
        return np.asarray(torch.as_tensor(boxes2D).detach().cpu(), np.float32).reshape(M, 4)
    out = np.zeros((M, 4), np.float32)
    for i in range(M):                      # detectron2 Boxes: my_loader.py:639
        b = boxes2D[i]
        t = b.tensor if hasattr(b, "tensor") else b
        out[i] = np.asarray(torch.as_tensor(t).detach().cpu(), np.float32).reshape(-1, 4)[0]
    return out


def depth2pointsrgbpm(depth, image, image_ori, calib, lidar, thing_classes, mask_image, classes,
                      scores, boxes2D, lidar_depth, file_idx, label_root=None, params=None,
                      plane=None, device="cuda:0", return_rows=False):
    """my_loader.py:502-617.  depth (Hd,Wd,1) float32 is thresholded in place like
    the reference (:507); returns all_points (n + P', 8) float64 (:611-617) and
    writes <label_root>/<file_idx>.txt (:545, :680).

    Sizes (hazard H11) follow the reference: the LiDAR side tests rounded pixels against
    params.bounds_hw (hard-coded 1600x900 at :526) and reads the masks there, whatever the
    size of `depth`; the pseudo side reads the masks at the depth pixels (:554).  Masks,
    depth and bounds are placed on one canvas large enough for all three (a mask smaller than
    the bounds reads as 0 outside -- the reference's PIL getpixel would raise there).
    More than DFU3D_MAX_INST (32) instances are processed in groups of 32 (instances are
    independent: the H3 row count K is the same for every instance)."""
    p = params or Params()
    depth = np.asarray(depth)
    Hd, Wd = depth.shape[0], depth.shape[1]
    depth[depth < p.depth_min] = 0                                   # :507
    lidar = np.ascontiguousarray(lidar, np.float32).reshape(-1, 4)
    masks = torch.as_tensor(mask_image).detach()
    M = int(masks.shape[0])
    Hm, Wm = (int(masks.shape[1]), int(masks.shape[2])) if M else (Hd, Wd)
    H = max(Hd, Hm, int(p.bounds_hw[0]))
    W = max(Wd, Wm, int(p.bounds_hw[1]))
    W = (W + 3) // 4 * 4                                             # float4 depth rows
    dev = torch.device(device)
    names_all = [thing_classes[int(c)] for c in classes]
    cls_all = [int(c) for c in classes]
    sc_all = [float(v) for v in np.asarray(torch.as_tensor(scores).cpu()).reshape(-1)[:M]] if M else []
    box_all = _boxes_array(boxes2D, M) if M else np.zeros((0, 4), np.float32)
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a)).to(dt).to(dev).contiguous()
    depth_c = torch.zeros((1, H, W), dtype=torch.float32, device=dev)
    depth_c[0, :Hd, :Wd] = t(depth.reshape(Hd, Wd), torch.float32)
    key = int(file_idx) if str(file_idx).isdigit() else 0
    G = 32                                                            # DFU3D_MAX_INST
    groups = [(g0, min(g0 + G, M)) for g0 in range(0, M, G)] or [(0, 0)]
    Mmax = min(G, max(8, (min(M, G) + 7) // 8 * 8))                  # few distinct engine shapes
    eng = _engine(p, H, W, Mmax, lidar.shape[0], device)
    rows_parts = []
    b = None
    for g0, g1 in groups:
        m = g1 - g0
        names, cls_idx = names_all[g0:g1], cls_all[g0:g1]
        mk = torch.zeros((1, Mmax, H, W), dtype=torch.uint8, device=dev)
        if m:
            mk[0, :m, :Hm, :Wm] = masks[g0:g1].to(dev).to(torch.uint8)   # np.uint8(mask), :525
        pad = Mmax - m
        b = ViewBatch(
            points=t(lidar if lidar.shape[0] else np.zeros((1, 4), np.float32), torch.float32),
            pt_off=t([0, lidar.shape[0]], torch.int32), view_frame=t([0], torch.int32),
            calib=t(calib.record()[None], torch.float32), masks=mk, n_inst=t([m], torch.int32),
            inst_class=t([cls_idx + [0] * pad], torch.int32),
            inst_is_car=t([[1 if n == "Car" else 0 for n in names] + [0] * pad], torch.int32),
            inst_r_lidar=t([[p.instance_radii(n)[0] for n in names] + [0.0] * pad], torch.float64),
            inst_r_pseudo=t([[p.instance_radii(n)[1] for n in names] + [0.0] * pad], torch.float64),
            inst_box=t(np.concatenate([box_all[g0:g1], np.zeros((pad, 4), np.float32)])[None], torch.float32),
            inst_score=t([sc_all[g0:g1] + [0.0] * pad], torch.float32),
            view_key=t([key], torch.int64), host_pt_off=np.array([0, lidar.shape[0]]),
            host_view_frame=np.array([0]), depth=depth_c,
            plane=None if plane is None else t(np.asarray(plane, np.float64).reshape(1, 4), torch.float64))
        rows, status = eng.run(b)
        if status:
            from ..stages import status_message
            raise RuntimeError("device status: " + status_message(status))
        r = rows.cpu().numpy()
        r[:, 1] += g0                                                 # instance index within the frame
        rows_parts.append(r)
    rows_h = np.concatenate(rows_parts, 0) if rows_parts else np.zeros((0, 24))
    if label_root is not None:
        os.makedirs(label_root, exist_ok=True)
        # the 10 nuScenes names index the row's class (my_loader.py:634, hazard H6)
        write_label_file(os.path.join(label_root, str(file_idx) + '.txt'), rows_h, NUSC_CLASSES)
    # virtual points: la_sampling2 representatives + rgb/3 (:539-542), lidar rows (:611-615)
    n_vox, vox_pix, xyz, st2 = eng.virtual_points(b)
    nv = int(n_vox[0].item())
    new_p1 = np.zeros((nv, 8))
    new_p1[:, 0:3] = xyz[0, :nv].cpu().numpy()
    if image is not None and nv:
        pix = vox_pix[0, :nv].cpu().numpy().astype(np.int64)          # canvas pixel -> (row, col) of the depth image
        new_p1[:, 4:7] = np.asarray(image)[pix // W, pix % W, :3] / 3
    new_p1[:, -1] = 1
    new_lidar = np.zeros((lidar.shape[0], 8))
    new_lidar[:, 0:4] = lidar[:, 0:4]
    new_lidar[:, 3] *= 10
    new_lidar[:, -1] = 2
    all_points = np.concatenate([new_lidar, new_p1], 0)
    return (all_points, rows_h) if return_rows else all_points
