"""KITTI-format frame files as the reference's converter lays them out
(tools/nuscenes2kitti.py:246-285): velodyne/%06d.bin (N,4) float32,
calib/%06d.txt with `%.12e` values, image_2/%06d.png, plus the two inputs the
foundation models would provide: seg_2/%06d.npz and depth_2/%06d.npy."""
import os

import numpy as np


def write_calib(path, P2, R0, V2C):
    """Keys and number format of tools/nuscenes2kitti.py:268-285."""
    def line(key, a):
        return key + ": " + " ".join("%.12e" % float(v) for v in np.asarray(a, np.float64).ravel()) + "\n"
    with open(path, "w") as f:
        for k in ("P0", "P1", "P2", "P3"):
            f.write(line(k, P2))
        f.write(line("R0_rect", R0))
        f.write(line("Tr_velo_to_cam", V2C))
        f.write(line("Tr_imu_to_velo", np.zeros((3, 4))))


def write_frame(root, idx, points, calib, image, masks, classes, scores, boxes, thing_classes, depth, compress=True):
    from PIL import Image
    name = "%06d" % idx if isinstance(idx, int) else str(idx)
    for d in ("velodyne", "calib", "image_2", "seg_2", "depth_2"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    np.asarray(points, np.float32).reshape(-1, 4).tofile(os.path.join(root, "velodyne", name + ".bin"))
    write_calib(os.path.join(root, "calib", name + ".txt"), calib.P2, calib.R0, calib.V2C)
    Image.fromarray(np.asarray(image, np.uint8)).save(os.path.join(root, "image_2", name + ".png"))
    # compress=False: 8 x 1.4 MB of masks per camera stay raw -- 4x the bytes on disk, but the readers of
    # dfu3d_amd.pipeline then only copy (the inflate is what bounds the batched labeler otherwise)
    (np.savez_compressed if compress else np.savez)(os.path.join(root, "seg_2", name + ".npz"), masks=np.asarray(masks, np.uint8),
                        classes=np.asarray(classes, np.int64), scores=np.asarray(scores, np.float32),
                        boxes=np.asarray(boxes, np.float32), thing_classes=np.array(thing_classes))
    np.save(os.path.join(root, "depth_2", name + ".npy"), np.asarray(depth, np.float32))
    return name
