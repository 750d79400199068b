"""GPU: the reference-facing surface (dfu3d_amd.penet.*) against the reference's
own outputs (golden G5/G7) and the oracle: depth2pointsrgbpm incl. the
virtual-point array (a16), LShapeFitting.fitting, save_depth_as_points and the
main.py CLI on a KITTI-format directory."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import penet_oracle as O

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from scene_image import pattern_image  # noqa: E402

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


class _Boxes:            # stand-in for detectron2 Boxes (my_loader.py:639)
    def __init__(self, row):
        self.tensor = torch.tensor(np.asarray(row, np.float32).reshape(1, 4))


@pytest.mark.parametrize("tag", ["dense", "dense2"])
def test_depth2pointsrgbpm_reproduces_reference(golden_dir, tmp_path, tag):
    _need_gpu()
    from dfu3d_amd.penet.calibration_kitti import Calibration
    from dfu3d_amd.penet.my_loader import depth2pointsrgbpm
    from dfu3d_amd.labels import read_label_file
    from dfu3d_amd.params import NUSC_CLASSES
    g = np.load(os.path.join(golden_dir, "g7_%s.npz" % tag))
    cpath = tmp_path / "c.txt"
    cpath.write_bytes(bytes(g["calib_text"]))
    calib = Calibration(str(cpath))
    lidar = g["lidar_all"][g["fov"]]                       # vis_utils.py:152-154 done by the caller
    depth = g["depth"].copy().reshape(900, 1600, 1)
    masks = torch.from_numpy(g["masks"].astype(np.float32))
    boxes = [_Boxes(b) for b in g["boxes"]]
    out = depth2pointsrgbpm(depth, pattern_image(), pattern_image(), calib, lidar, NUSC_CLASSES, masks,
                            g["classes"], np.ones(len(boxes), np.float32), boxes, None, "000700",
                            label_root=str(tmp_path / "lab"), plane=g["plane"])
    # label file == the reference's rows
    objs = read_label_file(str(tmp_path / "lab" / "000700.txt"))
    ref = [r.split(" ") for r in bytes(g["rows"]).decode().strip().split("\n")]
    assert len(objs) == len(ref)
    for o, w in zip(objs, ref):
        assert o.cls_type == w[0] and len(o.src.strip().split(" ")) == 15
        got = np.array([o.alpha, *o.box2d, o.h, o.w, o.l, *[float(v) for v in o.src.split(" ")[11:14]], o.ry])
        np.testing.assert_allclose(got, np.array(w[3:], float), rtol=1e-6, atol=1e-6)
    # virtual points (my_loader.py:611-617) == the reference's array
    assert out.shape == g["all_points"].shape
    np.testing.assert_allclose(out, g["all_points"], rtol=1e-12, atol=1e-9)
    # the in-place depth threshold of :507 is mirrored
    assert (depth[(depth > 0)] >= np.float32(0.001)).all()


def test_lshape_fitting_mirror_matches_reference(golden_dir):
    _need_gpu()
    from dfu3d_amd.penet.rectangle_fitting import LShapeFitting
    g = np.load(os.path.join(golden_dir, "g5_lshape.npz"))
    for name in [k[3:] for k in g.files if k.startswith("xy_")]:
        xy = g["xy_" + name]
        rects, idsets = LShapeFitting().fitting(xy[:, 0], xy[:, 1])
        assert len(idsets) == int(g["nclu_" + name]), name
        lab = np.full((xy.shape[0],), -1, np.int64)
        for k, ids in enumerate(idsets):
            lab[sorted(ids)] = k
        assert np.array_equal(lab, g["lab_" + name]), name
        abc = g["abc_" + name]
        assert len(rects) == abc.shape[0]
        for k, r in enumerate(rects):
            np.testing.assert_allclose(np.array([r.a, r.b, r.c]), abc[k], rtol=1e-9, atol=1e-9, err_msg=name)


def test_cli_on_kitti_directory_matches_oracle(tmp_path):
    _need_gpu()
    from dfu3d_amd import synth, kitti_io
    from dfu3d_amd.labels import read_label_file
    from dfu3d_amd.params import NUSC_CLASSES, Params
    from dfu3d_amd.penet import main as cli
    H, W, M = 225, 400, 6
    root = str(tmp_path / "kitti")
    exp = {}
    for f in range(2):
        s = synth.make_scene(60 + f, H=H, W=W, M=M, cams=1, dense=True, k_min=16, k_max=20)
        n = int(s.n_inst[0])
        img = pattern_image(H, W)
        kitti_io.write_frame(root, f, s.points.numpy(), s.calibs[0], img, s.masks[0][:n].numpy(),
                             s.inst_class[0][:n].numpy(), s.inst_score[0][:n].numpy(),
                             s.inst_box[0][:n].numpy(), NUSC_CLASSES, s.depth[0].numpy())
        # oracle on exactly what the files hold (calib goes through the %.12e text)
        oc = O.Calibration(os.path.join(root, "calib", "%06d.txt" % f))
        lid, _ = O.fov_filter(s.points.numpy(), oc, (H, W))
        op = O.Params(bounds_hw=(H, W), fov_hw=(H, W))
        res = O.depth2pointsrgbpm(s.depth[0].numpy().copy()[:, :, None], img, oc, lid, O.NUSC_CLASSES,
                                  s.masks[0][:n].numpy().astype(np.float32), s.inst_class[0][:n].numpy(),
                                  s.inst_box[0][:n].numpy(), op, plane_key=f, want_points=True)
        exp[f] = res
    rc = cli.main(["--command", "evaluate", "--detpath", root, "--conf_files", "x.yaml",
                   "--overrides", "WEIGHT", "none.pt"])
    assert rc == 0
    for f in range(2):
        objs = read_label_file(os.path.join(root, "label_2", "%06d.txt" % f))
        rows = exp[f].rows
        assert len(objs) == len(rows)
        for o, r in zip(objs, rows):
            assert o.cls_type == r.name
            got = np.array([o.alpha, *o.box2d, o.h, o.w, o.l, *[float(v) for v in o.src.split(" ")[11:14]], o.ry])
            np.testing.assert_allclose(got, r.as_vector(), rtol=1e-6, atol=1e-6)
        vp = np.load(os.path.join(root, "velodyne_depth", "%06d.npy" % f))
        assert vp.dtype == np.float16 and vp.shape == exp[f].all_points.shape
        assert np.array_equal(vp, exp[f].all_points.astype(np.float16))
    # a second run with --skip-existing leaves the files alone
    assert cli.main(["--detpath", root, "--skip-existing", "--conf_files", "x.yaml"]) == 0
    # the one-view-at-a-time path (save_depth_as_points) writes the same files as the batched one
    import shutil
    keep = {}
    for f in range(2):
        keep[f] = (open(os.path.join(root, "label_2", "%06d.txt" % f)).read(),
                   np.load(os.path.join(root, "velodyne_depth", "%06d.npy" % f)))
    shutil.rmtree(os.path.join(root, "label_2")); shutil.rmtree(os.path.join(root, "velodyne_depth"))
    assert cli.main(["--detpath", root, "--batch-frames", "0", "--conf_files", "x.yaml"]) == 0
    for f in range(2):
        assert open(os.path.join(root, "label_2", "%06d.txt" % f)).read() == keep[f][0]
        assert np.array_equal(np.load(os.path.join(root, "velodyne_depth", "%06d.npy" % f)), keep[f][1])


def test_batched_labeler_many_frames_partial_batch(tmp_path):
    """5 frames in batches of 4 (one padded batch), 2 streams, labels only."""
    _need_gpu()
    from dfu3d_amd import synth, kitti_io
    from dfu3d_amd.labels import read_label_file
    from dfu3d_amd.params import NUSC_CLASSES
    from dfu3d_amd.pipeline import BatchedLabeler
    H, W, M = 180, 320, 5
    root = str(tmp_path / "kitti")
    exp = {}
    for f in range(5):
        s = synth.make_scene(160 + f, H=H, W=W, M=M, cams=1, dense=True, k_min=14, k_max=18,
                             rings=(32, 24, 32, 16, 32)[f])
        n = int(s.n_inst[0])
        kitti_io.write_frame(root, f, s.points.numpy(), s.calibs[0], pattern_image(H, W), s.masks[0][:n].numpy(),
                             s.inst_class[0][:n].numpy(), s.inst_score[0][:n].numpy(),
                             s.inst_box[0][:n].numpy(), NUSC_CLASSES, s.depth[0].numpy())
        oc = O.Calibration(os.path.join(root, "calib", "%06d.txt" % f))
        lid, _ = O.fov_filter(s.points.numpy(), oc, (H, W))
        op = O.Params(bounds_hw=(H, W), fov_hw=(H, W))
        exp[f] = O.depth2pointsrgbpm(s.depth[0].numpy().copy()[:, :, None], None, oc, lid, O.NUSC_CLASSES,
                                     s.masks[0][:n].numpy().astype(np.float32), s.inst_class[0][:n].numpy(),
                                     s.inst_box[0][:n].numpy(), op, plane_key=f, want_points=False).rows
    lab = BatchedLabeler(batch_frames=4, lanes=2, workers=4)
    stats = lab.run(root, ["%06d" % f for f in range(5)], os.path.join(root, "label_2"))
    assert stats["frames"] == 5 and stats["boxes"] == sum(len(v) for v in exp.values())
    for f in range(5):
        objs = read_label_file(os.path.join(root, "label_2", "%06d.txt" % f))
        assert len(objs) == len(exp[f])
        for o, r in zip(objs, exp[f]):
            assert o.cls_type == r.name
            got = np.array([o.alpha, *o.box2d, o.h, o.w, o.l, *[float(v) for v in o.src.split(" ")[11:14]], o.ry])
            np.testing.assert_allclose(got, r.as_vector(), rtol=1e-6, atol=1e-6)


def test_integration_md_ctypes_example_runs_and_matches_oracle(monkeypatch):
    """The ctypes binding printed in INTEGRATION.md §2 is executed as written."""
    _need_gpu()
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    code = next(b for b in blocks if "def filter_and_fuse" in b)
    monkeypatch.chdir(root)
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    rng = np.random.default_rng(77)
    L = rng.normal(0, 1.5, (300, 3)); L[:5] += 40.0                 # a few isolated LiDAR points
    P = np.concatenate([L[rng.integers(5, 300, 900)] + rng.normal(0, 0.05, (900, 3)),
                        rng.normal(0, 1.5, (600, 3)), rng.uniform(-60, 60, (20, 3))])
    out = ns["filter_and_fuse"](torch.from_numpy(L).cuda(), torch.from_numpy(P).cuda()).cpu().numpy()
    exp, _, _ = O.instance_points("Car", L, P, O.Params())
    assert np.array_equal(out, exp)
