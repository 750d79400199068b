"""Build-container-only script: pins the f-2 box construction against the reference's own modules.

Imports (by file path, nothing is copied) pcdet/utils/calibration_kitti.py and pcdet/utils/object3d_kitti.py from
/root/reference, parses synthetic label lines with Object3d, converts the locations with Calibration.rect_to_lidar and
assembles gt_boxes_lidar with the expressions of pcdet/datasets/kitti/kitti_dataset.py:229-237.  Writes
tests/golden/g8_gtdb_boxes.npz (inputs + outputs, a few hundred bytes).
"""
import importlib.util
import os
import sys
import tempfile

import numpy as np

REF = "/root/reference/pcdet/utils"
HERE = os.path.dirname(os.path.abspath(__file__))


def load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    calib_mod, obj_mod = load("calibration_kitti"), load("object3d_kitti")
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from dfu3d_amd import kitti_io, synth
    rng = np.random.default_rng(88)
    cal = synth.make_calibration(35.0, 900, 1600, rng)
    tmp = tempfile.mkdtemp()
    cpath = os.path.join(tmp, "c.txt")
    kitti_io.write_calib(cpath, cal.P2, cal.R0, cal.V2C)
    lines = []
    names = ["Car", "Pedestrian", "Truck", "Car", "Barrier", "Car"]
    for i, nm in enumerate(names):
        v = rng.uniform(-1, 1, 16)
        box = [float(np.round(300 + 200 * v[0])), float(np.round(200 + 100 * v[1]))]
        box += [box[0] + float(np.round(20 + 80 * abs(v[2]))), box[1] + float(np.round(10 + 50 * abs(v[3])))]
        row = [nm, "0", "0", repr(float(v[4]))] + [repr(b) for b in box] + \
              [repr(float(1.5 + v[5])), repr(float(1.8 + 0.5 * v[6])), repr(float(4.0 + 2 * v[7])),
               repr(float(20 * v[8])), repr(float(1.5 + 0.3 * v[9])), repr(float(25 + 20 * v[10])), repr(float(3.1 * v[11]))]
        lines.append(" ".join(row))
    lpath = os.path.join(tmp, "l.txt")
    with open(lpath, "w") as f:
        f.write("\n".join(lines) + "\n")
    calib = calib_mod.Calibration(cpath)
    objs = obj_mod.get_objects_from_label(lpath)
    loc = np.concatenate([o.loc.reshape(1, 3) for o in objs], axis=0)
    dims = np.array([[o.l, o.h, o.w] for o in objs])
    rots = np.array([o.ry for o in objs])
    loc_lidar = calib.rect_to_lidar(loc)                                   # kitti_dataset.py:229
    l, h, w = dims[:, 0:1], dims[:, 1:2], dims[:, 2:3]
    loc_lidar[:, 2] += h[:, 0] / 2
    gt = np.concatenate([loc_lidar, l, w, h, -(np.pi / 2 + rots[..., np.newaxis])], axis=1)
    np.savez(os.path.join(HERE, "g8_gtdb_boxes.npz"), calib_text=open(cpath).read(), label_text=open(lpath).read(),
             gt_boxes_lidar=gt, loc_lidar_dtype=str(loc_lidar.dtype), level=np.array([o.level for o in objs]))
    print("wrote g8_gtdb_boxes.npz", gt.shape, loc_lidar.dtype, gt.dtype)


if __name__ == "__main__":
    main()
