"""KITTI label rows: writer for the 15-field rows the reference emits
(tools/PENet/dataloaders/my_loader.py:680) and a parser with the contract of
OpenPCDet's Object3d (pcdet/utils/object3d_kitti.py:20-37), used by tests to
check that what we write is what the detector's loader reads."""
from typing import Iterable, List

import numpy as np

from .params import NUSC_CLASSES

# column layout of an engine row (include/dfu3d.h, dfu3d_lshape_fit)
C_VIEW, C_INST, C_CLUSTER, C_CLASS, C_ALPHA = 0, 1, 2, 3, 4
C_BBOX, C_H, C_W, C_L, C_X, C_Y, C_Z, C_RY, C_SCORE, C_NPTS = 5, 9, 10, 11, 12, 13, 14, 15, 16, 17
C_THETA, C_RECT_C, C_ROOT = 18, 19, 23


def format_row(row, thing_classes=NUSC_CLASSES) -> str:
    """type trunc occ alpha x1 y1 x2 y2 h w l x y z ry  (my_loader.py:680).

    The reference writes Python str() of NumPy scalars: float32 for the 2-D box,
    float64 elsewhere (hazard H17); repr() round-trips both exactly."""
    r = np.asarray(row, np.float64)
    f = [thing_classes[int(r[C_CLASS])], '0', '0', repr(float(r[C_ALPHA]))]
    f += [repr(float(np.float32(v))) for v in r[C_BBOX:C_BBOX + 4]]
    f += [repr(float(r[c])) for c in (C_H, C_W, C_L, C_X, C_Y, C_Z, C_RY)]
    return ' '.join(f)


def write_label_file(path, rows: Iterable, thing_classes=NUSC_CLASSES) -> int:
    """One file per (frame, camera) view; an empty file is legal
    (pcdet/datasets/kitti/kitti_dataset.py:212,243-261)."""
    n = 0
    with open(path, 'w') as f:
        for r in rows:
            f.write(format_row(r, thing_classes) + '\n')
            n += 1
    return n


class LabelObject:
    """Same fields as pcdet/utils/object3d_kitti.py:Object3d.__init__."""

    def __init__(self, line: str):
        label = line.strip().split(' ')
        self.src = line
        self.cls_type = label[0]
        self.truncation = float(label[1])
        self.occlusion = float(label[2])
        self.alpha = float(label[3])
        self.box2d = np.array([float(v) for v in label[4:8]], dtype=np.float32)
        self.h, self.w, self.l = float(label[8]), float(label[9]), float(label[10])
        self.loc = np.array([float(v) for v in label[11:14]], dtype=np.float32)
        self.ry = float(label[14])
        self.score = float(label[15]) if len(label) == 16 else -1.0


def read_label_file(path) -> List[LabelObject]:
    with open(path) as f:
        return [LabelObject(l) for l in f.readlines() if l.strip()]
