"""Known answers for the f-2 oracle (points in boxes, LiDAR boxes from labels) -- CPU only."""
import numpy as np

from oracle import gtdb_oracle as G


def test_axis_aligned_box_membership():
    box = np.array([[1.0, 2.0, 0.5, 4.0, 2.0, 1.0, 0.0]])
    pts = np.array([[1.0, 2.0, 0.5],      # centre
                    [3.0, 2.0, 0.5],      # on the +x face (|lx| = dx/2 < dx/2 + margin)
                    [3.009, 2.0, 0.5],    # inside the margin
                    [3.011, 2.0, 0.5],    # outside the margin
                    [1.0, 3.02, 0.5],     # outside in y
                    [1.0, 2.0, 1.0],      # on the top face: |z-cz| = dz/2 is NOT > dz/2 -> inside
                    [1.0, 2.0, 1.001]], np.float32)
    got = G.points_in_boxes_cpu(pts, box)[0]
    assert got.tolist() == [1, 1, 1, 0, 0, 1, 0]


def test_rotated_box_membership():
    # heading 90 deg: the box's long side (dx = 4) lies along world y
    box = np.array([[0.0, 0.0, 0.0, 4.0, 1.0, 2.0, np.pi / 2]])
    pts = np.array([[0.0, 1.9, 0.0], [1.9, 0.0, 0.0], [0.4, -1.5, 0.9], [0.6, 0.0, 0.0]], np.float32)
    assert G.points_in_boxes_cpu(pts, box)[0].tolist() == [1, 0, 1, 0]


def test_gt_points_are_relative_to_the_float64_centre():
    pts = np.array([[10.123456, -3.5, 0.25, 0.7], [50.0, 0.0, 0.0, 0.1]], np.float32)
    box = np.array([10.1, -3.4, 0.2, 1.0, 1.0, 1.0, 0.3])
    ind = G.points_in_boxes_cpu(pts, box[None])[0]
    g = G.gt_points_of_box(pts, box, ind)
    assert g.shape == (1, 4) and g.dtype == np.float32
    assert g[0, 0] == np.float32(np.float64(pts[0, 0]) - 10.1) and g[0, 3] == pts[0, 3]


def test_obj_level():
    assert G.kitti_obj_level([0, 0, 10, 45], 0.0, 0) == 0
    assert G.kitti_obj_level([0, 0, 10, 30], 0.0, 0) == 1
    assert G.kitti_obj_level([0, 0, 10, 30], 0.4, 2) == 2
    assert G.kitti_obj_level([0, 0, 10, 10], 0.0, 0) == -1


def test_g8_boxes_from_labels_match_the_reference(tmp_path):
    """gt_boxes_lidar built by the reference's own Calibration / Object3d (tests/golden/capture_gtdb_golden.py) ==
    the oracle's and the product's construction, bit for bit."""
    import os
    from dfu3d_amd.calibration import Calibration
    from dfu3d_amd.labels import read_label_file
    from dfu3d_amd.pcdet_kitti.gt_database import annotations_from_label, rect_to_lidar
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g8_gtdb_boxes.npz"))
    cpath, lpath = str(tmp_path / "c.txt"), str(tmp_path / "l.txt")
    open(cpath, "w").write(str(g["calib_text"]))
    open(lpath, "w").write(str(g["label_text"]))
    calib = Calibration(cpath)
    objs = read_label_file(lpath)
    ann = annotations_from_label(objs, calib)
    assert ann["gt_boxes_lidar"].dtype == np.float64
    assert np.array_equal(ann["gt_boxes_lidar"], g["gt_boxes_lidar"])
    assert np.array_equal(ann["difficulty"], g["level"])
    ob = G.boxes_lidar_from_camera(np.stack([o.loc for o in objs]), [[o.l, o.h, o.w] for o in objs],
                                   [o.ry for o in objs], lambda p: rect_to_lidar(calib, p))
    assert np.array_equal(ob, g["gt_boxes_lidar"])
