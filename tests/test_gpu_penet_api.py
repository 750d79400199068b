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
    # per-frame log of the run: one line per labelled frame with its global index and box count
    import json
    log = os.path.join(root, "label_2", "frames.rank0.jsonl")
    recs = [json.loads(l) for l in open(log)]
    assert [r["frame"] for r in recs] == ["000000", "000001"] and [r["frame_idx"] for r in recs] == [0, 1]
    assert [r["boxes"] for r in recs] == [len(exp[0].rows), len(exp[1].rows)]
    # a second run with --skip-existing leaves the files alone (and logs nothing: nothing was labelled)
    assert cli.main(["--detpath", root, "--skip-existing", "--conf_files", "x.yaml"]) == 0
    assert len(open(log).readlines()) == 2
    # the done-list alone is enough for a restart: label file gone, frame listed -> not redone
    saved = open(os.path.join(root, "label_2", "000001.txt")).read()
    os.remove(os.path.join(root, "label_2", "000001.txt"))
    assert cli.main(["--detpath", root, "--resume", "--conf_files", "x.yaml"]) == 0
    assert not os.path.exists(os.path.join(root, "label_2", "000001.txt"))
    with open(os.path.join(root, "label_2", "000001.txt"), "w") as fh:
        fh.write(saved)
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
    # --keep-going: a frame with an unreadable depth file is logged and skipped, the other one is labelled
    shutil.rmtree(os.path.join(root, "label_2"))
    with open(os.path.join(root, "depth_2", "000000.npy"), "wb") as fh:
        fh.write(b"not a numpy file")
    assert cli.main(["--detpath", root, "--batch-frames", "0", "--keep-going", "--conf_files", "x.yaml"]) == 0
    recs = [json.loads(l) for l in open(os.path.join(root, "label_2", "frames.rank0.jsonl"))]
    assert [("error" in r, r["frame"]) for r in recs] == [(False, "000001"), (True, "000000")]
    assert open(os.path.join(root, "label_2", "000001.txt")).read() == keep[1][0]


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


# ------------------------------------------------------------------ golden G3: hazards H3 / H11 against the reference
def _check_label_file(path, ref_rows_text):
    from dfu3d_amd.labels import read_label_file
    objs = read_label_file(path)
    ref = [r.split(" ") for r in ref_rows_text.strip().split("\n")]
    assert len(objs) == len(ref)
    for o, w in zip(objs, ref):
        assert o.cls_type == w[0] and len(o.src.strip().split(" ")) == 15
        got = np.array([o.alpha, *o.box2d, o.h, o.w, o.l, *[float(v) for v in o.src.split(" ")[11:14]], o.ry])
        np.testing.assert_allclose(got, np.array(w[3:], float), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("kind", ["crop", "border"])
def test_g3_depth2pointsrgbpm_reproduces_reference(golden_dir, tmp_path, kind):
    """The mirror against the reference's own rows / virtual points for the MISALIGNED scene (K < n_ag, H3) and for
    the shipped size mix: FOV filter + depth on [:352,:1216], masks and the bounds test on 900x1600 (H11)."""
    _need_gpu()
    from dfu3d_amd.penet.calibration_kitti import Calibration
    from dfu3d_amd.penet.my_loader import depth2pointsrgbpm
    from dfu3d_amd.params import NUSC_CLASSES
    g = np.load(os.path.join(golden_dir, "g3_%s.npz" % kind))
    cpath = tmp_path / "c.txt"
    cpath.write_bytes(bytes(g["calib_text"]))
    calib = Calibration(str(cpath))
    lidar = g["lidar_all"][g["fov"]]                       # vis_utils.py:152-154 done by the caller
    dh, dw = g["depth"].shape
    depth = g["depth"].copy().reshape(dh, dw, 1)
    masks = torch.from_numpy(g["masks"].astype(np.float32))
    assert tuple(masks.shape[1:]) == (900, 1600)
    boxes = [_Boxes(b) for b in g["boxes"]]
    img = pattern_image()[:dh, :dw]
    out = depth2pointsrgbpm(depth, img, img, calib, lidar, NUSC_CLASSES, masks, g["classes"],
                            np.ones(len(boxes), np.float32), boxes, None, "000310",
                            label_root=str(tmp_path / "lab"), plane=g["plane"])
    _check_label_file(str(tmp_path / "lab" / "000310.txt"), bytes(g["rows"]).decode())
    assert out.shape == g["all_points"].shape
    np.testing.assert_allclose(out, g["all_points"], rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("kind", ["crop", "border"])
def test_g3_engine_with_fov_size_different_from_bounds(golden_dir, tmp_path, kind):
    """PseudoBoxEngine doing the FOV pre-filter itself (apply_fov) with fov_hw != bounds_hw ('crop') and with the
    border points that round out of bounds ('border'): rows == the reference's label rows."""
    _need_gpu()
    from dfu3d_amd.calibration import Calibration
    from dfu3d_amd.engine import PseudoBoxEngine, ViewBatch
    from dfu3d_amd.labels import write_label_file
    from dfu3d_amd.params import NUSC_CLASSES, Params
    g = np.load(os.path.join(golden_dir, "g3_%s.npz" % kind))
    cpath = tmp_path / "c.txt"
    cpath.write_bytes(bytes(g["calib_text"]))
    calib = Calibration(str(cpath))
    fov_hw = tuple(int(x) for x in g["fov_hw"])
    p = Params(fov_hw=fov_hw)                              # bounds stay 900x1600 (my_loader.py:526)
    H, W, M = 900, 1600, 8
    dev = "cuda:0"
    lid = g["lidar_all"]
    m = g["masks"].shape[0]
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a)).to(dt).to(dev).contiguous()
    mk = torch.zeros((1, M, H, W), dtype=torch.uint8, device=dev)
    mk[0, :m] = t(g["masks"], torch.uint8)
    depth = torch.zeros((1, H, W), dtype=torch.float32, device=dev)
    dh, dw = g["depth"].shape
    depth[0, :dh, :dw] = t(g["depth"], torch.float32)
    names = [NUSC_CLASSES[int(c)] for c in g["classes"]]
    pad = M - m
    b = ViewBatch(points=t(lid, torch.float32), pt_off=t([0, lid.shape[0]], torch.int32), view_frame=t([0], torch.int32),
                  calib=t(calib.record()[None], torch.float32), masks=mk, n_inst=t([m], torch.int32),
                  inst_class=t([[int(c) for c in g["classes"]] + [0] * pad], torch.int32),
                  inst_is_car=t([[1 if n == "Car" else 0 for n in names] + [0] * pad], torch.int32),
                  inst_r_lidar=t([[p.instance_radii(n)[0] for n in names] + [0.0] * pad], torch.float64),
                  inst_r_pseudo=t([[p.instance_radii(n)[1] for n in names] + [0.0] * pad], torch.float64),
                  inst_box=t(np.concatenate([g["boxes"], np.zeros((pad, 4), np.float32)])[None], torch.float32),
                  inst_score=t([[1.0] * m + [0.0] * pad], torch.float32), view_key=t([0], torch.int64),
                  host_pt_off=np.array([0, lid.shape[0]]), host_view_frame=np.array([0]), depth=depth,
                  plane=t(np.asarray(g["plane"], np.float64).reshape(1, 4), torch.float64))
    for fmt in ("bytes", "packed"):
        if fmt == "packed":
            b.pack_masks()
        for chain in (False, True):
            eng = PseudoBoxEngine(p, H, W, M, lid.shape[0], views_per_chunk=1, dense=True, cap_vox=1 << 18,
                                  pool_per_view=1 << 19, rows_per_view=256, apply_fov=True, chain=chain)
            rows, status = eng.run(b)
            assert status == 0
            # n_ag / K of the reference (hazard H3)
            if not chain:
                assert int(eng.n_fov[0]) == int(g["fov"].sum())
                assert int(eng.n_ag[0]) == int(g["n_ag"]) and int(eng.K[0]) == int(g["K"])
            path = str(tmp_path / ("%s_%d.txt" % (fmt, chain)))
            write_label_file(path, rows.cpu().numpy(), NUSC_CLASSES)
            _check_label_file(path, bytes(g["rows"]).decode())
            del eng
            torch.cuda.empty_cache()


def test_module_swap_of_integration_md(golden_dir, tmp_path, monkeypatch):
    """the reference's vis_utils.py:136-166 with ONLY its two imports swapped
    (calibration_kitti, depth2pointsrgbpm) -- host-side FOV pre-filter through the swapped Calibration's
    lidar_to_rect / rect_to_img, then the swapped depth2pointsrgbpm -- reproduces the reference's label rows."""
    _need_gpu()
    from dfu3d_amd.penet import calibration_kitti                      # swapped import 1
    from dfu3d_amd.penet.my_loader import depth2pointsrgbpm            # swapped import 2
    from dfu3d_amd.params import NUSC_CLASSES
    g = np.load(os.path.join(golden_dir, "g7_dense.npz"))
    root = tmp_path / "kitti"
    for d in ("calib", "velodyne"):
        (root / d).mkdir(parents=True)
    (root / "calib" / "000700.txt").write_bytes(bytes(g["calib_text"]))
    g["lidar_all"].astype(np.float32).tofile(str(root / "velodyne" / "000700.bin"))

    def get_fov_flag(pts_rect, img_shape, calib):                      # vis_utils.py:108-123, as the reference has it
        pts_img, pts_rect_depth = calib.rect_to_img(pts_rect)
        f1 = np.logical_and(pts_img[:, 0] >= 0, pts_img[:, 0] < img_shape[1])
        f2 = np.logical_and(pts_img[:, 1] >= 0, pts_img[:, 1] < img_shape[0])
        return np.logical_and(np.logical_and(f1, f2), pts_rect_depth >= 0)

    def save_depth_as_points_stub(depth, idx, root_path):              # the shape of vis_utils.py:136-166
        file_idx = str(idx).zfill(6)
        calib = calibration_kitti.Calibration(os.path.join(root_path, 'calib', file_idx + '.txt'))
        lidar = np.fromfile(os.path.join(root_path, 'velodyne', file_idx + '.bin'), dtype=np.float32).reshape(-1, 4)
        image = pattern_image()
        masks = torch.from_numpy(g["masks"].astype(np.float32))        # SegSeem stand-in
        boxes = [_Boxes(b_) for b_ in g["boxes"]]
        pts_rect = calib.lidar_to_rect(lidar[:, 0:3])
        lidar = lidar[get_fov_flag(pts_rect, image.shape, calib)]
        return depth2pointsrgbpm(depth.reshape(900, 1600, 1), image, image, calib, lidar, NUSC_CLASSES, masks,
                                 g["classes"], np.ones(len(boxes), np.float32), boxes, None, file_idx,
                                 label_root=os.path.join(root_path, "label_2"), plane=g["plane"])

    out = save_depth_as_points_stub(g["depth"].copy(), 700, str(root))
    _check_label_file(str(root / "label_2" / "000700.txt"), bytes(g["rows"]).decode())
    np.testing.assert_allclose(out, g["all_points"], rtol=1e-12, atol=1e-9)


def test_more_than_32_instances_are_all_labelled(tmp_path):
    """A frame with 40 instances (DFU3D_MAX_INST = 32): the mirror processes them in groups and labels every one,
    in instance order, like the reference's loop over mask_image.shape[0] (my_loader.py:547)."""
    _need_gpu()
    from dfu3d_amd import synth
    from dfu3d_amd.calibration import Calibration
    from dfu3d_amd.penet.my_loader import depth2pointsrgbpm
    from dfu3d_amd.params import NUSC_CLASSES, Params
    H, W, M = 180, 320, 8
    s = synth.make_scene(77, H=H, W=W, M=M, cams=1, dense=True, k_min=16, k_max=20)
    n = int(s.n_inst[0])
    assert n >= 2
    reps = (40 + n - 1) // n
    idx = np.tile(np.arange(n), reps)[:40]
    masks = s.masks[0][:n].numpy()[idx].astype(np.float32)
    classes = s.inst_class[0][:n].numpy()[idx]
    boxes = s.inst_box[0][:n].numpy()[idx]
    cal = s.calibs[0]
    calib = Calibration({"P2": cal.P2, "R0": cal.R0, "Tr_velo2cam": cal.V2C})
    oc = O.Calibration({"P2": cal.P2, "R0": cal.R0, "Tr_velo2cam": cal.V2C})
    lid, _ = O.fov_filter(s.points.numpy(), oc, (H, W))
    p = Params(bounds_hw=(H, W), fov_hw=(H, W))
    op = O.Params(bounds_hw=(H, W), fov_hw=(H, W))
    res = O.depth2pointsrgbpm(s.depth[0].numpy().copy()[:, :, None], None, oc, lid, O.NUSC_CLASSES, masks, classes,
                              boxes, op, plane_key=5, want_points=False)
    _, rows = depth2pointsrgbpm(s.depth[0].numpy().copy()[:, :, None], None, None, calib, lid, NUSC_CLASSES,
                                torch.from_numpy(masks), classes, np.ones(40, np.float32), boxes, None, "000005",
                                label_root=str(tmp_path / "lab"), params=p, return_rows=True)
    assert len(res.rows) == rows.shape[0] and rows.shape[0] > 0
    assert int(rows[:, 1].max()) >= 32                      # instances beyond the 32nd produced boxes
    for got, r in zip(rows, res.rows):
        assert (int(got[1]), int(got[2]), int(got[3])) == (r.inst, r.cluster, r.cls)
        np.testing.assert_allclose(got[4:16], r.as_vector(), rtol=1e-6, atol=1e-6)


def test_cli_batched_path_labels_every_instance_of_a_frame_with_40(tmp_path, capsys):
    """`--batch-frames 4` (a batch view holds DFU3D_MAX_INST = 32 instances): a frame with 40 instances must come out
    exactly as the oracle -- and as `--batch-frames 0` -- labels it (my_loader.py:547 loops over every mask), the
    others of the batch untouched; the run says so instead of cutting the frame short."""
    _need_gpu()
    import shutil
    from dfu3d_amd import synth, kitti_io
    from dfu3d_amd.labels import read_label_file
    from dfu3d_amd.params import NUSC_CLASSES
    from dfu3d_amd.penet import main as cli
    H, W, M = 180, 320, 8
    root = str(tmp_path / "kitti")
    exp = {}
    for f in range(3):
        s = synth.make_scene(77 + f, H=H, W=W, M=M, cams=1, dense=True, k_min=16, k_max=20)
        n = int(s.n_inst[0])
        assert n >= 2
        idx = np.tile(np.arange(n), (40 + n - 1) // n)[:40] if f == 1 else np.arange(n)      # frame 1: 40 instances
        masks, classes = s.masks[0][:n].numpy()[idx], s.inst_class[0][:n].numpy()[idx]
        boxes, scores = s.inst_box[0][:n].numpy()[idx], np.full(len(idx), 0.9, np.float32)
        img = pattern_image(H, W)
        kitti_io.write_frame(root, f, s.points.numpy(), s.calibs[0], img, masks, classes, scores, boxes, NUSC_CLASSES,
                             s.depth[0].numpy())
        oc = O.Calibration(os.path.join(root, "calib", "%06d.txt" % f))
        lid, _ = O.fov_filter(s.points.numpy(), oc, (H, W))
        op = O.Params(bounds_hw=(H, W), fov_hw=(H, W))
        exp[f] = O.depth2pointsrgbpm(s.depth[0].numpy().copy()[:, :, None], img, oc, lid, O.NUSC_CLASSES,
                                     masks.astype(np.float32), classes, boxes, op, plane_key=f, want_points=False)
    assert max(r.inst for r in exp[1].rows) >= 32                      # the case is real: boxes beyond the 32nd instance
    for procs in ("0", "1"):                  # reader threads | forked reader processes (the m_total path of reader_pool)
        if procs == "1":                      # labels only: no image_2/ is needed, in the fallback for the 40-instance frame either
            shutil.move(os.path.join(root, "image_2"), os.path.join(root, "image_2_away"))
        assert cli.main(["--detpath", root, "--batch-frames", "4", "--reader-procs", procs, "--conf_files", "x.yaml"]
                        + (["--no-virtual-points"] if procs == "1" else [])) == 0
        assert "000001: 40 instances" in capsys.readouterr().err
        for f in range(3):
            objs = read_label_file(os.path.join(root, "label_2", "%06d.txt" % f))
            rows = exp[f].rows
            assert len(objs) == len(rows), (f, len(objs), len(rows))
            for o, r in zip(objs, rows):
                assert o.cls_type == r.name
                got = np.array([o.alpha, *o.box2d, o.h, o.w, o.l, *[float(v) for v in o.src.split(" ")[11:14]], o.ry])
                np.testing.assert_allclose(got, r.as_vector(), rtol=1e-6, atol=1e-6)
        batched = {f: open(os.path.join(root, "label_2", "%06d.txt" % f)).read() for f in range(3)}
        shutil.rmtree(os.path.join(root, "label_2"))
    shutil.move(os.path.join(root, "image_2_away"), os.path.join(root, "image_2"))
    assert cli.main(["--detpath", root, "--batch-frames", "0", "--conf_files", "x.yaml"]) == 0
    for f in range(3):
        assert open(os.path.join(root, "label_2", "%06d.txt" % f)).read() == batched[f]


def test_two_rank_cli_run_equals_the_single_rank_run(tmp_path):
    """BASELINE configs[2] in small: the split sharded over the ranks of one node.  Two FRESH child processes
    (`python -m torch.distributed.run --nproc-per-node 2 -m dfu3d_amd.penet.main ... --dist-backend gloo
    --single-device`: both on this one GPU, rows gathered over gloo) label a 12-frame directory; the label files must be
    byte-identical to a single-process run's, frames must be split r::2 (per-rank logs), and rank 0's manifest must
    account for every box of both ranks."""
    _need_gpu()
    import json
    import subprocess
    import sys
    from dfu3d_amd import synth, kitti_io
    from dfu3d_amd.params import NUSC_CLASSES
    H, W, M, N = 180, 320, 6, 12
    root = str(tmp_path / "kitti")
    img = pattern_image(H, W)
    for f in range(N):
        s = synth.make_scene(300 + f, H=H, W=W, M=M, cams=1, dense=True, k_min=14, k_max=18)
        n = int(s.n_inst[0])
        kitti_io.write_frame(root, f, s.points.numpy(), s.calibs[0], img, s.masks[0][:n].numpy(), s.inst_class[0][:n].numpy(),
                             s.inst_score[0][:n].numpy(), s.inst_box[0][:n].numpy(), NUSC_CLASSES, s.depth[0].numpy())
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = ["-m", "dfu3d_amd.penet.main", "--command", "evaluate", "--detpath", root, "--no-virtual-points",
            "--batch-frames", "4", "--reader-procs", "2"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")

    def run(cmd):
        r = subprocess.run(cmd, cwd=repo, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (cmd, r.stdout[-1500:], r.stderr[-3000:])
    one, two = os.path.join(root, "label_one"), os.path.join(root, "label_two")
    run([sys.executable] + base + ["--label-out", one])
    run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", "29531"] + base + ["--label-out", two, "--dist-backend", "gloo", "--single-device"])
    names = ["%06d.txt" % f for f in range(N)]
    for nm in names:
        a, b = open(os.path.join(one, nm)).read(), open(os.path.join(two, nm)).read()
        assert a == b, nm
    n_boxes = sum(len(open(os.path.join(one, nm)).read().splitlines()) for nm in names)
    assert n_boxes > 20
    man1, man2 = json.load(open(os.path.join(one, "manifest.json"))), json.load(open(os.path.join(two, "manifest.json")))
    assert man1["frames"] == man2["frames"] == N and man1["boxes"] == man2["boxes"] == n_boxes
    assert man2["world_size"] == 2 and sum(man2["boxes_per_rank"]) == n_boxes and min(man2["boxes_per_rank"]) > 0
    for r in (0, 1):                                        # the reference's DistributedSampler interleave: rank r has frames r::2
        recs = [json.loads(l) for l in open(os.path.join(two, "frames.rank%d.jsonl" % r))]
        assert sorted(x["frame_idx"] for x in recs) == list(range(r, N, 2))
