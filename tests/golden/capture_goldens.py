#!/usr/bin/env python3
"""Capture golden vectors from the reference's own Python modules.

Runs ONLY in the build container (needs /root/reference); the GPU box never
sees the reference.  Output: tests/golden/*.npz + label rows -- data only
(inputs and the reference's outputs), no reference source text.

What is imported unmodified from /root/reference/tools/PENet:
  dataloaders/calibration_kitti.py, rectangle_fitting/rectangle_fitting.py,
  dataloaders/my_loader.py (estimate_plane, above_plane, distance_to_plane,
  get_fov_flag, to_sphere_coords, BallQuery, GenerateAnns, calc_rect_contour,
  depth2pointsrgbpm).
Third-party packages the reference imports but this image lacks (open3d, cv2,
skimage, spconv) are provided as import-time stand-ins; the two that do
arithmetic on the path (spconv point-to-voxel, Open3D remove_radius_outlier)
delegate to the oracle's restatement, so G7 pins the reference's
*orchestration, ordering and formatting*, not those leaves.

Usage: python tests/golden/capture_goldens.py
"""
import io
import os
import sys
import tempfile
import types

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/tools/PENet"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

import numpy as np
import torch

from oracle import penet_oracle as O
sys.path.insert(0, HERE)
from scene_image import pattern_image  # noqa: E402

RECORD = {"v3d": []}


# ---------------------------------------------------------------- stand-ins
def _install_stubs():
    o3d = types.ModuleType("open3d")

    class _PC:
        def __init__(self):
            self.points = np.zeros((0, 3))

        def remove_radius_outlier(self, nb_points, radius):
            pts = np.asarray(self.points, np.float64).reshape(-1, 3)
            idx = O.radius_outlier(pts, nb_points, radius)
            out = _PC()
            out.points = pts[idx]
            return out, list(idx)

        def uniform_down_sample(self, every_k_points):
            out = _PC()
            out.points = np.asarray(self.points)[::every_k_points]
            return out

    def _v3d(a):
        a = np.array(a, np.float64).reshape(-1, 3)
        RECORD["v3d"].append(a.copy())
        return a

    o3d.geometry = types.SimpleNamespace(PointCloud=_PC)
    o3d.utility = types.SimpleNamespace(Vector3dVector=_v3d)
    sys.modules["open3d"] = o3d
    sys.modules["cv2"] = types.ModuleType("cv2")
    sk = types.ModuleType("skimage")
    sk.io = types.ModuleType("skimage.io")
    sys.modules["skimage"] = sk
    sys.modules["skimage.io"] = sk.io

    spconv = types.ModuleType("spconv")
    sp_utils = types.ModuleType("spconv.utils")
    sp_torch = types.ModuleType("spconv.pytorch")

    class VoxelGeneratorV2:
        """Stand-in for spconv's point-to-voxel leaf (delegates to the oracle)."""

        def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels):
            self.p = O.Params(
                vsize=tuple(float(np.float32(v)) for v in voxel_size),
                vrange_min=tuple(float(np.float32(v)) for v in point_cloud_range[:3]),
                max_points_per_voxel=max_num_points, max_voxels=max_voxels)

        def generate(self, points):
            import ctypes
            pts = np.ascontiguousarray(points, np.float64)
            m, F = pts.shape
            grid = np.array(self.p.vgrid, np.int32)
            tab = np.full((int(grid.prod()),), -1, np.int32)
            vop = np.empty((m,), np.int32)
            cnt = np.empty((max(m, 1),), np.int32)
            cell = np.empty((max(m, 1),), np.int64)
            nv = O._lib().orc_voxelize(
                O._p(pts), ctypes.c_int64(F), ctypes.c_int64(m),
                O._p(np.array(self.p.vsize)), O._p(np.array(self.p.vrange_min)),
                O._p(grid), ctypes.c_int32(self.p.max_points_per_voxel),
                ctypes.c_int32(self.p.max_voxels), O._p(tab), O._p(vop),
                O._p(cnt), O._p(cell))
            voxels = np.zeros((nv, self.p.max_points_per_voxel, F), pts.dtype)
            fill = np.zeros((nv,), np.int64)
            for i in range(m):
                v = vop[i]
                if v >= 0:
                    voxels[v, fill[v]] = pts[i]
                    fill[v] += 1
            return voxels, cell[:nv], cnt[:nv].copy()

    sp_utils.VoxelGeneratorV2 = VoxelGeneratorV2
    spconv.utils = sp_utils
    spconv.pytorch = sp_torch
    sys.modules["spconv"] = spconv
    sys.modules["spconv.utils"] = sp_utils
    sys.modules["spconv.pytorch"] = sp_torch


_install_stubs()
from dataloaders import calibration_kitti as ref_calib          # noqa: E402
from rectangle_fitting.rectangle_fitting import LShapeFitting   # noqa: E402
from dataloaders import my_loader as ref_loader                  # noqa: E402


# ---------------------------------------------------------------- helpers
def write_calib(path, P2, R0, V2C):
    def line(key, a):
        return key + ": " + " ".join("%.12e" % v for v in np.asarray(a).ravel()) + "\n"
    with open(path, "w") as f:
        f.write(line("P0", P2))
        f.write(line("P1", P2))
        f.write(line("P2", P2))
        f.write(line("P3", P2))
        f.write(line("R0_rect", R0))
        f.write(line("Tr_velo_to_cam", V2C))
        f.write(line("Tr_imu_to_velo", np.zeros((3, 4))))


def make_calib(rng, yaw_deg=0.0, r0_noise=0.0):
    fx = 1266.417203046554
    P2 = np.array([[fx, 0, 816.2670197447984, 0.0],
                   [0, fx, 491.50706579294757, 0.0],
                   [0, 0, 1, 0.0]], np.float64)
    if r0_noise:
        P2[0, 3] = 45.3
        P2[1, 3] = -2.1
        P2[2, 3] = 0.004
    yaw = np.deg2rad(yaw_deg)
    Rz = np.array([[np.cos(yaw), np.sin(yaw), 0], [-np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]])
    A = np.array([[0, -1, 0], [0, 0, -1], [1, 0, 0]], np.float64)
    R = A @ Rz
    t = np.array([0.012, -0.33, -0.55]) + rng.normal(0, 0.01, 3)
    V2C = np.hstack([R, t.reshape(3, 1)])
    R0 = np.eye(3)
    if r0_noise:
        w = rng.normal(0, r0_noise, 3)
        K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        R0 = np.eye(3) + K + K @ K / 2
    return P2, R0, V2C


class _Boxes:
    def __init__(self, row):
        self.tensor = torch.tensor(np.asarray(row, np.float32).reshape(1, 4))


def lshape(rng, cx, cy, L, Wd, yaw_deg, n, noise=0.02):
    """Points on the two visible edges of a rotated rectangle."""
    yaw = np.deg2rad(yaw_deg)
    k = n // 2
    e1 = np.stack([rng.uniform(-L / 2, L / 2, k), np.full(k, -Wd / 2)], 1)
    e2 = np.stack([np.full(n - k, -L / 2), rng.uniform(-Wd / 2, Wd / 2, n - k)], 1)
    p = np.vstack([e1, e2]) + rng.normal(0, noise, (n, 2))
    R = np.array([[np.cos(yaw), -np.sin(yaw)], [np.sin(yaw), np.cos(yaw)]])
    p = p @ R.T + np.array([cx, cy])
    return p[rng.permutation(n)]


# ---------------------------------------------------------------- G1 calib
def g1(tmp):
    rng = np.random.default_rng(101)
    out = {}
    for tag, yaw, r0n in (("a", 0.0, 0.0), ("b", 55.0, 0.004)):
        P2, R0, V2C = make_calib(rng, yaw, r0n)
        path = os.path.join(tmp, "g1%s.txt" % tag)
        write_calib(path, P2, R0, V2C)
        c = ref_calib.Calibration(path)
        pts32 = np.concatenate([rng.normal(0, 25, (250, 3)),
                                rng.normal(0, 1, (6, 3))]).astype(np.float32)
        pts64 = rng.normal(0, 25, (16, 3))
        rect32 = c.lidar_to_rect(pts32)
        img32, dep32 = c.rect_to_img(rect32)
        img_b, dep_b = c.lidar_to_img(pts32)
        u = rng.integers(0, 1600, 256).astype(np.int64)
        v = rng.integers(0, 900, 256).astype(np.int64)
        d = rng.uniform(0.5, 80, 256).astype(np.float32)
        prect = c.img_to_rect(u, v, d)
        plid = c.rect_to_lidar(prect)
        fov = ref_loader.get_fov_flag(rect32, (900, 1600), c)
        out.update({
            tag + "_text": np.frombuffer(open(path, "rb").read(), np.uint8),
            tag + "_P2": c.P2, tag + "_R0": c.R0, tag + "_V2C": c.V2C,
            tag + "_scal": np.array([c.cu, c.cv, c.fu, c.fv, c.tx, c.ty], np.float32),
            tag + "_pts32": pts32, tag + "_pts64": pts64,
            tag + "_rect32": rect32, tag + "_rect64": c.lidar_to_rect(pts64),
            tag + "_img32": img32, tag + "_dep32": dep32,
            tag + "_img_b": img_b, tag + "_dep_b": dep_b,
            tag + "_u": u, tag + "_v": v, tag + "_d": d,
            tag + "_prect": prect, tag + "_plid": plid, tag + "_fov": fov,
        })
    np.savez_compressed(os.path.join(HERE, "g1_calib.npz"), **out)


# ---------------------------------------------------------------- G2 plane
def synth_sweep(rng, n=6000, tilt=(0.01, -0.02), h=-1.84):
    xy = rng.uniform(-60, 60, (n, 2))
    z = tilt[0] * xy[:, 0] + tilt[1] * xy[:, 1] + h + rng.normal(0, 0.02, n)
    obj = rng.random(n) < 0.12
    z[obj] += rng.uniform(0.1, 9.5, obj.sum())
    return np.concatenate([xy, z[:, None], rng.uniform(0, 255, (n, 1))], 1).astype(np.float32)


def g2():
    out = {}
    for k, (tilt, h) in enumerate((((0.0, 0.0), -1.84), ((0.015, -0.02), -1.7),
                                   ((-0.03, 0.01), -2.0))):
        rng = np.random.default_rng(200 + k)
        pts = synth_sweep(rng, 6000, tilt, h)
        np.random.seed(0)                       # hazard H1
        plane = ref_loader.estimate_plane(pts[:, :3])
        mask = ref_loader.above_plane(pts[:, :3], plane)
        dist = ref_loader.distance_to_plane(pts[:, :3], plane, directional=True)
        out.update({"pts%d" % k: pts, "plane%d" % k: plane,
                    "mask%d" % k: mask, "dist%d" % k: dist})
    np.savez_compressed(os.path.join(HERE, "g2_plane.npz"), **out)


# ---------------------------------------------------------------- G4 ballquery
def g4():
    rng = np.random.default_rng(400)
    out = {}
    for k, (n1, n2) in enumerate(((300, 40), (50, 1), (7, 120))):
        p2 = rng.normal(0, 1.0, (n2, 3))
        p1 = np.concatenate([p2[rng.integers(0, n2, n1 // 2)] + rng.normal(0, 0.06, (n1 // 2, 3)),
                             rng.normal(0, 1.0, (n1 - n1 // 2, 3))])
        _, min_dis, _, mask = ref_loader.BallQuery(torch.from_numpy(p1), torch.from_numpy(p2), 0.1)
        out.update({"p1_%d" % k: p1, "p2_%d" % k: p2, "mask%d" % k: mask.numpy(),
                    "mind%d" % k: min_dis.numpy()})
    np.savez_compressed(os.path.join(HERE, "g4_ballquery.npz"), **out)


# ---------------------------------------------------------------- G5 / G6
def g5_cases():
    rng = np.random.default_rng(500)
    cases = {}
    # analytic axis-aligned rectangle outline (dense, noise-free)
    t = np.linspace(-2.3, 2.3, 24)
    s = np.linspace(-0.95, 0.95, 11)
    rect = np.vstack([np.stack([t, np.full_like(t, -0.95)], 1),
                      np.stack([np.full_like(s, -2.3), s], 1)]) + np.array([12.0, 3.0])
    cases["rect0"] = rect
    for yaw in (7.0, 33.4, 61.0, 88.6):
        cases["L%s" % str(yaw).replace(".", "p")] = lshape(rng, 15.0, -4.0, 4.6, 1.9, yaw, 60)
    a = lshape(rng, 10.0, 5.0, 4.5, 1.8, 20.0, 30)
    b = lshape(rng, 22.0, -6.0, 4.2, 1.9, 70.0, 30)
    c = lshape(rng, 35.0, 8.0, 7.0, 2.5, 45.0, 24)
    two = np.empty((60, 2))
    two[0::2], two[1::2] = a, b
    cases["two_interleaved"] = two
    three = np.vstack([c[:5], a[:10], b[:12], c[5:], a[10:], b[12:]])
    cases["three_mixed"] = three
    cases["n1"] = np.array([[5.0, 1.0]])
    cases["n2"] = np.array([[5.0, 1.0], [6.5, 1.4]])
    cases["n3"] = np.array([[5.0, 1.0], [6.5, 1.4], [5.4, 2.9]])
    cases["far_pair"] = np.array([[5.0, 1.0], [8.004, 1.0], [30.0, 0.0], [33.02, 0.0]])
    cases["chain"] = np.stack([np.arange(12) * 2.9 + 3.0, np.zeros(12)], 1)[rng.permutation(12)]
    return cases


def g5_g6(tmp):
    rng = np.random.default_rng(600)
    P2, R0, V2C = make_calib(rng, 0.0, 0.003)
    cpath = os.path.join(tmp, "000007.txt")
    write_calib(cpath, P2, R0, V2C)
    out = {"calib_text": np.frombuffer(open(cpath, "rb").read(), np.uint8)}
    rows_txt = []
    names = O.NUSC_CLASSES
    for ci, (name, xy) in enumerate(g5_cases().items()):
        fit = LShapeFitting()
        rects, idsets = fit.fitting(xy[:, 0], xy[:, 1])
        assert rects is not None
        out["xy_" + name] = xy
        out["nclu_" + name] = np.array(len(idsets))
        lab = np.full((xy.shape[0],), -1, np.int64)
        for k, ids in enumerate(idsets):
            lab[sorted(ids)] = k
        out["lab_" + name] = lab
        out["abc_" + name] = np.array([[r.a, r.b, r.c] for r in rects], np.float64)
        # G6: GenerateAnns rows for two classes (aspect-skip applies to Car)
        z = rng.uniform(-1.6, 0.4, xy.shape[0])
        pts = np.concatenate([xy, z[:, None]], 1)
        out["z_" + name] = z
        for seem_name, cls_idx in (("Car", 0), (names[(ci % 9) + 1], (ci % 9) + 1)):
            f = io.StringIO()
            box = np.array([570.0 + ci, 177.5, 666.25, 247.0], np.float32)
            ref_loader.GenerateAnns(seem_name, pts, cls_idx, None, _Boxes(box),
                                    tmp + os.sep, "000007", f, [])
            for line in f.getvalue().splitlines():
                rows_txt.append("%s|%s|%d|%s" % (name, seem_name, cls_idx, line))
    np.savez_compressed(os.path.join(HERE, "g5_lshape.npz"), **out)
    with open(os.path.join(HERE, "g6_rows.txt"), "w") as f:
        f.write("\n".join(rows_txt) + "\n")


# ---------------------------------------------------------------- G7 whole function
def g7_scene(seed, dense):
    """Small scene at full 900x1600 geometry: a few box-like objects."""
    rng = np.random.default_rng(seed)
    H, W = 900, 1600
    P2, R0, V2C = make_calib(rng, 0.0, 0.0)
    # LiDAR: ground + three objects in front of the camera
    n_g = 2500
    ang = rng.uniform(-0.62, 0.62, n_g)
    rad = rng.uniform(3.0, 60.0, n_g)
    ground = np.stack([rad * np.cos(ang), rad * np.sin(ang),
                       -1.84 + rng.normal(0, 0.02, n_g)], 1)
    objs = [("Car", 0, (12.0, 1.5), (4.6, 1.9, 1.6), 25.0),
            ("Pedestrian", 8, (27.0, -2.5), (0.6, 0.6, 1.75), 0.0),
            ("Truck", 1, (24.0, -5.0), (7.5, 2.6, 2.0), 70.0)]
    pts = [ground]
    obj_pts = []
    for _, _, (cx, cy), (L, Wd, Hh), yaw in objs:
        m = int(70 * (10.0 / cx) ** 1.2) + 12
        e = lshape(rng, cx, cy, L, Wd, yaw, m, 0.015)
        z = rng.uniform(-1.7, -1.84 + Hh, m)
        p = np.concatenate([e, z[:, None]], 1)
        obj_pts.append(p)
        pts.append(p)
    # a few points that land within half a pixel of the right/bottom border (H3)
    lidar = np.concatenate(pts).astype(np.float32)
    lidar = lidar[rng.permutation(lidar.shape[0])]
    lidar = np.concatenate([lidar, rng.uniform(0, 255, (lidar.shape[0], 1)).astype(np.float32)], 1)
    calib = ref_calib.Calibration({"P2": P2.astype(np.float32), "R0": R0.astype(np.float32),
                                   "Tr_velo2cam": V2C.astype(np.float32),
                                   "P3": P2.astype(np.float32)})
    # masks: projected bounding rectangle of each object's points (+margin)
    masks = np.zeros((len(objs), H, W), np.float32)
    boxes = np.zeros((len(objs), 4), np.float32)
    depth = np.zeros((H, W, 1), np.float32)
    for k, p in enumerate(obj_pts):
        img, dep = calib.lidar_to_img(p.astype(np.float32))
        x0, y0 = np.floor(img.min(0)).astype(int) - 3
        x1, y1 = np.ceil(img.max(0)).astype(int) + 3
        x0, y0, x1, y1 = max(x0, 0), max(y0, 0), min(x1, W - 1), min(y1, H - 1)
        masks[k, y0:y1 + 1, x0:x1 + 1] = 1.0
        boxes[k] = (x0, y0, x1, y1)
        if dense:
            # fronto-parallel-ish depth patch over the object (plus noise) and
            # some ground around it
            yy, xx = np.mgrid[y0:y1 + 1, x0:x1 + 1]
            dd = float(np.median(dep)) + 0.002 * (xx - x0) + rng.normal(0, 0.03, xx.shape)
            depth[y0:y1 + 1, x0:x1 + 1, 0] = dd.astype(np.float32)
    if dense:
        gy0 = 640
        yy, xx = np.mgrid[gy0:760, 650:950]
        # ground depth from the pinhole model: y_cam = 1.51 m below the camera
        dd = 1.51 * P2[1, 1] / (yy - P2[1, 2] + 1e-3)
        dd *= 1 + rng.normal(0, 0.01, dd.shape)
        cur = depth[gy0:760, 650:950, 0]
        depth[gy0:760, 650:950, 0] = np.where(cur > 0, cur, dd.astype(np.float32))
        sp = rng.random(depth.shape) < 0.002
        depth[sp & (depth > 0)] *= 0.3
    image = pattern_image(H, W)
    classes = np.array([o[1] for o in objs], np.int64)
    return dict(P2=P2.astype(np.float32), R0=R0.astype(np.float32), V2C=V2C.astype(np.float32),
                lidar=lidar, masks=masks, boxes=boxes, depth=depth, image=image, classes=classes)


def run_reference(sc, stem, tmp, fov_hw):
    """vis_utils.py:152-154 (FOV pre-filter with `fov_hw`) + the reference's
    depth2pointsrgbpm on scene `sc`; records what it hands to Open3D and to
    GenerateAnns, the label text and the returned virtual points."""
    cpath = os.path.join(tmp, "%s.txt" % stem)
    write_calib(cpath, sc["P2"], sc["R0"], sc["V2C"])
    calib = ref_calib.Calibration(cpath)
    flag = ref_loader.get_fov_flag(calib.lidar_to_rect(sc["lidar"][:, :3]), fov_hw, calib)
    lidar = sc["lidar"][flag]
    label_buf = io.StringIO()
    label_buf.close = lambda: None
    ann_inputs = []
    orig_ann = ref_loader.GenerateAnns

    def rec_ann(cls_name, pts, *a, **k):
        ann_inputs.append(np.array(pts, np.float64).reshape(-1, 3))
        return orig_ann(cls_name, pts, *a, **k)

    def fake_open(path, mode="r", *a, **k):
        if "w" in mode:
            return label_buf
        return open(os.path.join(tmp, os.path.basename(path)), mode, *a, **k)

    class _CalibShim:
        @staticmethod
        def Calibration(path):
            return ref_calib.Calibration(os.path.join(tmp, os.path.basename(path)))

    ref_loader.open = fake_open
    ref_loader.GenerateAnns = rec_ann
    ref_loader.calibration_kitti = _CalibShim
    RECORD["v3d"].clear()
    np.random.seed(0)
    plane = ref_loader.estimate_plane(lidar[:, :3])
    np.random.seed(0)                     # same RNG state inside the call
    depth_in = sc["depth"].copy()
    boxes = [_Boxes(b) for b in sc["boxes"]]
    all_points = ref_loader.depth2pointsrgbpm(
        depth_in, sc["image"], sc["image"], calib, lidar, O.NUSC_CLASSES,
        torch.from_numpy(sc["masks"]), sc["classes"],
        np.ones(len(boxes), np.float32), boxes, None, stem)
    ref_loader.GenerateAnns = orig_ann
    ref_loader.calibration_kitti = ref_calib
    del ref_loader.open
    rows = label_buf.getvalue()
    out = dict(P2=sc["P2"], R0=sc["R0"], V2C=sc["V2C"],
               calib_text=np.frombuffer(open(cpath, "rb").read(), np.uint8),
               lidar_all=sc["lidar"], fov=flag,
               masks=sc["masks"].astype(np.uint8), boxes=sc["boxes"],
               depth=sc["depth"][:, :, 0], classes=sc["classes"],
               plane=plane, all_points=all_points,
               rows=np.frombuffer(rows.encode(), np.uint8),
               n_v3d=np.array(len(RECORD["v3d"])), n_ann=np.array(len(ann_inputs)))
    for i, a in enumerate(RECORD["v3d"]):
        out["v3d_%d" % i] = a
    for i, a in enumerate(ann_inputs):
        out["ann_%d" % i] = a
    return out, calib, lidar, plane, rows


def g7(tmp):
    for tag, seed, dense in (("dense", 700, True), ("dense2", 701, True)):
        sc = g7_scene(seed, dense)
        # vis_utils.py:152-154 FOV pre-filter with the canonical (900,1600)
        out, _, _, _, rows = run_reference(sc, "%06d" % seed, tmp, (900, 1600))
        np.savez_compressed(os.path.join(HERE, "g7_%s.npz" % tag), **out)
        print(tag, "rows:\n" + rows)


# ---------------------------------------------------------------- G3 label inheritance (hazards H3 / H11)
def plant_pixel(calib, rng, u_lo, u_hi, v_lo, v_hi, d_lo, d_hi, fov_hw):
    """A float32 LiDAR point whose float32 projection (the reference's lidar_to_img)
    lands in [u_lo,u_hi) x [v_lo,v_hi) and passes the FOV pre-filter of `fov_hw`."""
    for _ in range(2000):
        u, v, d = rng.uniform(u_lo, u_hi), rng.uniform(v_lo, v_hi), rng.uniform(d_lo, d_hi)
        p = calib.rect_to_lidar(calib.img_to_rect(np.array([u]), np.array([v]), np.array([d])))
        p32 = p.astype(np.float32)
        img, _ = calib.lidar_to_img(p32)
        ok = ref_loader.get_fov_flag(calib.lidar_to_rect(p32), fov_hw, calib)[0]
        if ok and u_lo <= img[0, 0] < u_hi and v_lo <= img[0, 1] < v_hi:
            return p32[0]
    raise RuntimeError("could not plant a point")


def g3_scene(kind):
    """Two vehicles + ground; `kind`:
      'crop'   -- the shipped size mix (hazard H11): KITTI-like intrinsics, FOV pre-filter and
                  depth on the [:352,:1216] crop (vis_utils.py:147,153,161), masks and the bounds
                  test on the hard-coded 1600x900 canvas (my_loader.py:526); points are planted
                  whose pixel ROUNDS to column 1216 / row 352 -- outside the crop, inside the
                  bounds, looked up in the mask there;
      'border' -- canonical 900x1600 everywhere with points planted at u in [1599.5,1600) and
                  v in [899.5,900): they pass the unrounded FOV test, round to 1600 / 900 and fail
                  the bounds test, so K < n_ag and every later row inherits the label of the
                  NEXT in-bounds pixel (hazard H3)."""
    crop = kind == "crop"
    rng = np.random.default_rng(310 if crop else 311)
    H, W = 900, 1600
    fov_hw = (352, 1216) if crop else (900, 1600)
    P2, R0, V2C = make_calib(rng, 0.0, 0.0)
    if crop:
        P2[0, 0] = P2[1, 1] = 721.5377
        P2[0, 2], P2[1, 2] = 609.5593, 172.854
    calib = ref_calib.Calibration({"P2": P2.astype(np.float32), "R0": R0.astype(np.float32),
                                   "Tr_velo2cam": V2C.astype(np.float32), "P3": P2.astype(np.float32)})
    n_g = 1800
    ang = rng.uniform(-0.75, 0.75, n_g)
    rad = rng.uniform(3.0, 60.0, n_g)
    ground = np.stack([rad * np.cos(ang), rad * np.sin(ang), -1.84 + rng.normal(0, 0.02, n_g)], 1)
    objs = [("Car", 0, (11.0, 1.2), (4.6, 1.9, 1.6), 20.0),
            ("Bus", 3, (21.0, -4.0), (9.0, 2.8, 2.6), 65.0)]
    pts, obj_pts = [ground], []
    for _, _, (cx, cy), (L, Wd, Hh), yaw in objs:
        m = int(70 * (10.0 / cx) ** 1.2) + 12
        e = lshape(rng, cx, cy, L, Wd, yaw, m, 0.015)
        z = rng.uniform(-1.7, -1.84 + Hh, m)
        p = np.concatenate([e, z[:, None]], 1)
        obj_pts.append(p)
        pts.append(p)
    fh, fw = fov_hw
    planted = []
    for _ in range(6):          # right border: rounds to column fw
        planted.append(plant_pixel(calib, rng, fw - 0.5, fw, 40.0, fh - 40.0, 4.0, 18.0, fov_hw))
    for _ in range(6):          # bottom border: rounds to row fh (near, so that it is above the ground plane)
        planted.append(plant_pixel(calib, rng, 200.0, fw - 200.0, fh - 0.5, fh, 1.6, 3.2, fov_hw))
    planted = np.stack(planted).astype(np.float64)
    lidar = np.concatenate(pts + [planted]).astype(np.float32)
    lidar = lidar[rng.permutation(lidar.shape[0])]
    lidar = np.concatenate([lidar, rng.uniform(0, 255, (lidar.shape[0], 1)).astype(np.float32)], 1)
    masks = np.zeros((len(objs), H, W), np.float32)
    boxes = np.zeros((len(objs), 4), np.float32)
    dh, dw = fov_hw
    depth = np.zeros((dh, dw, 1), np.float32)
    for k, p in enumerate(obj_pts):
        img, dep = calib.lidar_to_img(p.astype(np.float32))
        x0, y0 = np.floor(img.min(0)).astype(int) - 3
        x1, y1 = np.ceil(img.max(0)).astype(int) + 3
        x0, y0, x1, y1 = max(x0, 0), max(y0, 0), min(x1, W - 1), min(y1, H - 1)
        masks[k, y0:y1 + 1, x0:x1 + 1] = 1.0
        boxes[k] = (x0, y0, x1, y1)
        xa, ya, xb, yb = x0, y0, min(x1, dw - 1), min(y1, dh - 1)
        if xb >= xa and yb >= ya:
            yy, xx = np.mgrid[ya:yb + 1, xa:xb + 1]
            dd = float(np.median(dep)) + 0.002 * (xx - xa) + rng.normal(0, 0.03, xx.shape)
            depth[ya:yb + 1, xa:xb + 1, 0] = dd.astype(np.float32)
    # the masks also cover the border strips the planted points round into
    masks[0, :, min(fw, W - 1) - 2:min(fw, W - 1) + 3] = 1.0
    masks[1, min(fh, H - 1) - 2:min(fh, H - 1) + 3, :] = 1.0
    image = pattern_image(H, W)[:dh, :dw]
    classes = np.array([o[1] for o in objs], np.int64)
    return dict(P2=P2.astype(np.float32), R0=R0.astype(np.float32), V2C=V2C.astype(np.float32),
                lidar=lidar, masks=masks, boxes=boxes, depth=depth, image=image, classes=classes), fov_hw


def g3(tmp):
    for kind in ("crop", "border"):
        sc, fov_hw = g3_scene(kind)
        out, calib, lidar, plane, rows = run_reference(sc, "0003%s" % ("10" if kind == "crop" else "11"), tmp, fov_hw)
        # n_ag / K of my_loader.py:512-527 with the reference's own functions
        lm = lidar[ref_loader.above_plane(lidar[:, :3], plane)]
        pts_img, _ = calib.lidar_to_img(lm[:, :3])
        r = np.round(pts_img)
        inb = (0 <= r[:, 0]) & (r[:, 0] < 1600) & (0 <= r[:, 1]) & (r[:, 1] < 900)
        n_ag, K = int(lm.shape[0]), int(min(inb.sum(), lm.shape[0]))
        beyond_crop = int(((r[:, 0] >= fov_hw[1]) | (r[:, 1] >= fov_hw[0])).sum())
        print("g3", kind, "n_ag", n_ag, "K", K, "rounded beyond the FOV size", beyond_crop)
        if kind == "border":
            assert K < n_ag, "the border scene must be misaligned (hazard H3)"
            assert (~inb)[: n_ag // 2].any(), "an out-of-bounds row must precede in-bounds rows"
        else:
            assert K == n_ag and beyond_crop >= 6
        out.update(n_ag=np.array(n_ag), K=np.array(K), fov_hw=np.array(fov_hw), beyond=np.array(beyond_crop))
        np.savez_compressed(os.path.join(HERE, "g3_%s.npz" % kind), **out)
        print(kind, "rows:\n" + rows)


def main():
    """python capture_goldens.py [g1 g2 g3 g4 g5 g7]   (default: all)"""
    tmp = tempfile.mkdtemp(prefix="dfu3d_gold_")
    want = set(sys.argv[1:]) or {"g1", "g2", "g3", "g4", "g5", "g7"}
    if "g1" in want:
        g1(tmp)
    if "g2" in want:
        g2()
    if "g3" in want:
        g3(tmp)
    if "g4" in want:
        g4()
    if "g5" in want:
        g5_g6(tmp)
    if "g7" in want:
        g7(tmp)
    print("goldens written to", HERE)


if __name__ == "__main__":
    main()
