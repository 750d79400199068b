"""f-2: points in boxes / ground-truth database on the GPU vs the oracle -- bit exact."""
import os
import pickle

import numpy as np
import pytest
import torch

from oracle import gtdb_oracle as G

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _scene(rng, n, nb):
    pts = np.zeros((n, 4), np.float32)
    pts[:, :2] = rng.uniform(-40, 40, (n, 2))
    pts[:, 2] = rng.uniform(-2.5, 1.5, n)
    pts[:, 3] = rng.random(n)
    boxes = np.zeros((nb, 7))
    boxes[:, :2] = rng.uniform(-35, 35, (nb, 2))
    boxes[:, 2] = rng.uniform(-1.5, 0.0, nb)
    boxes[:, 3:6] = rng.uniform(0.4, 12.0, (nb, 3))
    boxes[:, 6] = rng.uniform(-4, 4, nb)
    # points planted on faces / inside the margin band of the first boxes
    for b in range(min(nb, 4) if n >= 64 else 0):
        c, s = np.cos(boxes[b, 6]), np.sin(boxes[b, 6])
        for k, fx in enumerate((0.5, 0.5 + 0.009 / boxes[b, 3], 0.5 + 0.011 / boxes[b, 3])):
            lx, ly = fx * boxes[b, 3], 0.1 * boxes[b, 4]
            pts[10 * b + k, 0] = boxes[b, 0] + lx * c - ly * s
            pts[10 * b + k, 1] = boxes[b, 1] + lx * s + ly * c
            pts[10 * b + k, 2] = boxes[b, 2] + (0.5 * boxes[b, 5] if k == 0 else 0.0)
    return pts, boxes


def test_points_in_boxes_mask_matches_oracle():
    from dfu3d_amd.pcdet_kitti.roiaware_pool3d_utils import points_in_boxes_cpu
    rng = np.random.default_rng(5)
    for n, nb in ((0, 3), (7, 0), (1, 1), (5000, 17), (34720, 40)):
        pts, boxes = _scene(rng, max(n, 1), max(nb, 1))
        pts, boxes = pts[:n], boxes[:nb]
        got = points_in_boxes_cpu(pts[:, :3].copy(), boxes)
        exp = G.points_in_boxes_cpu(pts, boxes)
        assert got.dtype == np.int32 and got.shape == (nb, n)
        assert np.array_equal(got, exp), (n, nb, int((got != exp).sum()))
        if n >= 64 and nb:
            assert exp.sum() > 0


def test_gt_database_many_frames_matches_oracle():
    from dfu3d_amd import stages as st
    rng = np.random.default_rng(6)
    frames = [_scene(rng, n, nb) for n, nb in ((3000, 9), (1, 2), (20000, 31), (500, 0), (8000, 5))]
    off = np.zeros(len(frames) + 1, np.int64)
    off[1:] = np.cumsum([f[0].shape[0] for f in frames])
    pts = torch.from_numpy(np.concatenate([f[0] for f in frames])).to(DEV)
    boxes = np.concatenate([f[1] for f in frames])
    bf = np.repeat(np.arange(len(frames), dtype=np.int32), [f[1].shape[0] for f in frames])
    cap = int(off[-1]) * 2
    cnt, boff, idx, gt, status = st.gt_database(pts, torch.from_numpy(off.astype(np.int32)).to(DEV),
                                                torch.from_numpy(bf).to(DEV), torch.from_numpy(boxes).to(DEV), cap,
                                                host_pt_off=off)
    assert int(status.item()) == 0
    cnt, boff, idx, gt = cnt.cpu().numpy(), boff.cpu().numpy(), idx.cpu().numpy(), gt.cpu().numpy()
    k = 0
    for p, b in frames:
        ind = G.points_in_boxes_cpu(p, b)
        for i in range(b.shape[0]):
            exp_idx = np.nonzero(ind[i])[0]
            assert cnt[k] == len(exp_idx) and boff[k + 1] - boff[k] == len(exp_idx)
            assert np.array_equal(idx[boff[k]:boff[k + 1]], exp_idx)
            exp_pts = G.gt_points_of_box(p, b[i], ind[i])
            assert np.array_equal(gt[boff[k]:boff[k + 1]].view(np.uint32), exp_pts.view(np.uint32))   # bit exact
            k += 1
    # capacity overflow is flagged, counts stay exact
    c2, _, _, _, s2 = st.gt_database(pts, torch.from_numpy(off.astype(np.int32)).to(DEV), torch.from_numpy(bf).to(DEV),
                                     torch.from_numpy(boxes).to(DEV), 8, host_pt_off=off)
    assert int(s2.item()) & st.ST_POOL_OVERFLOW and np.array_equal(c2.cpu().numpy(), cnt)


def test_create_groundtruth_database_from_pseudo_labels(tmp_path):
    """Directory of frames + the label files the hot path wrote -> gt_database/*.bin + dbinfos, against the oracle."""
    from dfu3d_amd import kitti_io, synth
    from dfu3d_amd.calibration import Calibration
    from dfu3d_amd.labels import read_label_file
    from dfu3d_amd.params import NUSC_CLASSES
    from dfu3d_amd.pcdet_kitti.gt_database import create_groundtruth_database, rect_to_lidar
    from dfu3d_amd.penet import main as cli
    from tests.golden.scene_image import pattern_image
    H, W, M = 180, 320, 5
    root = str(tmp_path / "kitti")
    for f in range(3):
        s = synth.make_scene(300 + f, H=H, W=W, M=M, cams=1, dense=True, k_min=14, k_max=18)
        n = int(s.n_inst[0])
        kitti_io.write_frame(root, f, s.points.numpy(), s.calibs[0], pattern_image(H, W), s.masks[0][:n].numpy(),
                             s.inst_class[0][:n].numpy(), s.inst_score[0][:n].numpy(), s.inst_box[0][:n].numpy(),
                             NUSC_CLASSES, s.depth[0].numpy())
    assert cli.main(["--detpath", root, "--no-virtual-points", "--conf_files", "x.yaml"]) == 0
    ids = ["%06d" % f for f in range(3)]
    infos = create_groundtruth_database(root, ids, batch_frames=2)
    with open(os.path.join(root, "kitti_dbinfos_train.pkl"), "rb") as fh:
        assert set(pickle.load(fh).keys()) == set(infos.keys())
    total = 0
    for sid in ids:
        pts = np.fromfile(os.path.join(root, "velodyne", sid + ".bin"), np.float32).reshape(-1, 4)
        calib = Calibration(os.path.join(root, "calib", sid + ".txt"))
        objs = read_label_file(os.path.join(root, "label_2", sid + ".txt"))
        boxes = G.boxes_lidar_from_camera(np.stack([o.loc for o in objs]) if objs else np.zeros((0, 3)),
                                          [[o.l, o.h, o.w] for o in objs], [o.ry for o in objs],
                                          lambda p: rect_to_lidar(calib, p))
        ind = G.points_in_boxes_cpu(pts, boxes)
        for i, o in enumerate(objs):
            exp = G.gt_points_of_box(pts, boxes[i], ind[i])
            got = np.fromfile(os.path.join(root, "gt_database", "%s_%s_%d.bin" % (sid, o.cls_type, i)),
                              np.float32).reshape(-1, 4)
            assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), (sid, i)
            rec = [r for r in infos[o.cls_type] if r["image_idx"] == sid and r["gt_idx"] == i][0]
            assert rec["num_points_in_gt"] == exp.shape[0] and np.array_equal(rec["box3d_lidar"], boxes[i])
            assert rec["difficulty"] == G.kitti_obj_level(o.box2d, o.truncation, o.occlusion)
            total += exp.shape[0]
    assert total > 0
