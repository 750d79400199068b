"""Ground-truth database from KITTI label files, batched on the GPU.

Reference: pcdet/datasets/kitti/kitti_dataset.py:158-262 (get_infos: annotations, gt_boxes_lidar) and :284-331
(create_groundtruth_database: per box the points inside, relative to the centre, as
gt_database/<frame>_<class>_<i>.bin plus kitti_dbinfos_<split>.pkl).  The reference handles one frame and one
box at a time on the CPU; here `batch_frames` frames and all their boxes go through dfu3d_gt_database in one launch
chain and the files are written by background threads.
"""
import os
import pickle
from concurrent.futures import ThreadPoolExecutor
from typing import List

import numpy as np
import torch

from .. import stages as st
from ..calibration import Calibration, fma_chain_matmul_f32
from ..labels import read_label_file


def kitti_obj_level(obj):
    """pcdet/utils/object3d_kitti.py:39-53."""
    height = float(obj.box2d[3]) - float(obj.box2d[1]) + 1
    if height >= 40 and obj.truncation <= 0.15 and obj.occlusion <= 0:
        return 0
    if height >= 25 and obj.truncation <= 0.3 and obj.occlusion <= 1:
        return 1
    if height >= 25 and obj.truncation <= 0.5 and obj.occlusion <= 2:
        return 2
    return -1


def rect_to_lidar(calib: Calibration, pts_rect):
    """pcdet/utils/calibration_kitti.py:50-63 (float32 points -> float32 sgemm with inv((R0_ext @ V2C_ext)^T))."""
    p = np.asarray(pts_rect, np.float32).reshape(-1, 3)
    hom = np.hstack([p, np.ones((p.shape[0], 1), np.float32)])
    return fma_chain_matmul_f32(hom, calib.Minv)[:, 0:3]


def annotations_from_label(objs, calib: Calibration):
    """kitti_dataset.py:211-237 for one frame -> dict with name / bbox / score / difficulty / gt_boxes_lidar."""
    objs = list(objs)
    names = np.array([o.cls_type for o in objs])
    num_objects = len([o for o in objs if o.cls_type != 'DontCare'])
    ann = {
        'name': names,
        'bbox': np.concatenate([o.box2d.reshape(1, 4) for o in objs], 0) if objs else np.zeros((0, 4)),
        'score': np.array([o.score for o in objs]) if objs else np.zeros((0,)),
        'difficulty': np.array([kitti_obj_level(o) for o in objs], np.int32) if objs else np.zeros((0,), np.int32),
    }
    if num_objects > 0:
        loc = np.concatenate([o.loc.reshape(1, 3) for o in objs], 0)[:num_objects]           # float32
        dims = np.array([[o.l, o.h, o.w] for o in objs])[:num_objects]                         # lhw(camera), float64
        rots = np.array([o.ry for o in objs])[:num_objects]
        loc_lidar = rect_to_lidar(calib, loc)                                                  # float32
        l, h, w = dims[:, 0:1], dims[:, 1:2], dims[:, 2:3]
        loc_lidar[:, 2] += h[:, 0] / 2
        ann['gt_boxes_lidar'] = np.concatenate([loc_lidar, l, w, h, -(np.pi / 2 + rots[..., np.newaxis])], axis=1)
    else:
        ann['gt_boxes_lidar'] = np.zeros((0, 7))
    return ann


def create_groundtruth_database(root_path, sample_ids: List[str], label_dir=None, used_classes=None, split='train',
                                batch_frames=64, device="cuda:0", workers=8):
    """Writes <root>/gt_database[_<split>]/<id>_<name>_<i>.bin and <root>/kitti_dbinfos_<split>.pkl
    (kitti_dataset.py:284-331) for the frames `sample_ids`; labels from `label_dir` (default <root>/label_2),
    points from <root>/velodyne/<id>.bin, calibration from <root>/calib/<id>.txt.  Returns the db-info dict."""
    label_dir = label_dir or os.path.join(root_path, 'label_2')
    db_dir = os.path.join(root_path, 'gt_database' if split == 'train' else 'gt_database_%s' % split)
    os.makedirs(db_dir, exist_ok=True)
    all_db_infos = {}
    writers = ThreadPoolExecutor(workers)
    pending = []

    def read(sid):
        pts = np.fromfile(os.path.join(root_path, 'velodyne', sid + '.bin'), dtype=np.float32).reshape(-1, 4)
        calib = Calibration(os.path.join(root_path, 'calib', sid + '.txt'))
        lab = os.path.join(label_dir, sid + '.txt')
        objs = read_label_file(lab) if os.path.exists(lab) else []
        return sid, pts, annotations_from_label(objs, calib)

    def write(path, arr):
        with open(path, 'w') as f:                    # kitti_dataset.py:320-321
            arr.tofile(f)

    readers = ThreadPoolExecutor(workers)
    for b0 in range(0, len(sample_ids), batch_frames):
        frames = list(readers.map(read, sample_ids[b0:b0 + batch_frames]))
        sizes = [f[1].shape[0] for f in frames]
        off = np.zeros(len(frames) + 1, np.int64)
        off[1:] = np.cumsum(sizes)
        boxes = [f[2]['gt_boxes_lidar'] for f in frames]
        nb = [b.shape[0] for b in boxes]
        if sum(nb) == 0:
            continue
        pts_d = torch.from_numpy(np.concatenate([f[1] for f in frames], 0)).to(device)
        boxes_d = torch.from_numpy(np.concatenate(boxes, 0)).to(device=device, dtype=torch.float64)
        bf = torch.from_numpy(np.repeat(np.arange(len(frames), dtype=np.int32), nb)).to(device)
        cap = int(off[-1]) * 2 + 1024                 # boxes of one frame rarely share points
        cnt, boff, idx, gt, status = st.gt_database(pts_d, torch.from_numpy(off.astype(np.int32)).to(device), bf,
                                                    boxes_d, cap, host_pt_off=off)
        if int(status.item()):                        # overlapping boxes: size exactly and repeat
            cap = int(boff[-1].item()) + 1
            cnt, boff, idx, gt, status = st.gt_database(pts_d, torch.from_numpy(off.astype(np.int32)).to(device), bf,
                                                        boxes_d, cap, host_pt_off=off)
        cnt_h, off_h = cnt.cpu().numpy(), boff.cpu().numpy()
        gt_h = gt[:int(off_h[-1])].cpu().numpy()
        k = 0
        for (sid, _, ann), n_f in zip(frames, nb):
            for i in range(n_f):
                name = ann['name'][i]
                filename = '%s_%s_%d.bin' % (sid, name, i)
                g = gt_h[off_h[k]:off_h[k + 1]]
                pending.append(writers.submit(write, os.path.join(db_dir, filename), g))
                if (used_classes is None) or name in used_classes:
                    info = {'name': name, 'path': os.path.relpath(os.path.join(db_dir, filename), root_path),
                            'image_idx': sid, 'gt_idx': i, 'box3d_lidar': ann['gt_boxes_lidar'][i],
                            'num_points_in_gt': int(cnt_h[k]), 'difficulty': ann['difficulty'][i],
                            'bbox': ann['bbox'][i], 'score': ann['score'][i]}
                    all_db_infos.setdefault(name, []).append(info)
                k += 1
    for p in pending:
        p.result()
    writers.shutdown()
    readers.shutdown()
    with open(os.path.join(root_path, 'kitti_dbinfos_%s.pkl' % split), 'wb') as f:
        pickle.dump(all_db_infos, f)
    return all_db_infos
