"""Synthetic nuScenes-like scenes for tests and bench.

One scene = one 32-ring LiDAR sweep (34,720 rays by default) + C pinhole
cameras looking at a ground plane and a handful of oriented boxes; per camera a
z-buffer gives the dense depth map and exact instance silhouettes.  Everything
is generated with torch ops on the requested device (CPU for small tests, GPU
for bench batches); no dataset or network access is needed.  This module is
data plumbing, not part of the measured path.
"""
import math
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch

from .calibration import Calibration
from .params import NUSC_CLASSES, Params

# (class name, l, w, h)
BOX_TYPES = [("Car", 4.6, 1.9, 1.7), ("Truck", 7.0, 2.5, 3.0), ("Pedestrian", 0.7, 0.7, 1.8),
             ("Traffic_cone", 0.4, 0.4, 0.8), ("Bus", 10.5, 2.9, 3.3), ("Bicycle", 1.8, 0.6, 1.4)]
CAM_YAWS_DEG = [0.0, 55.0, -55.0, 110.0, -110.0, 180.0]
GROUND_Z = -1.84


@dataclass
class Scene:
    points: torch.Tensor          # (N,4) f32
    calibs: List[Calibration]     # C
    depth: torch.Tensor           # (C,H,W) f32
    masks: torch.Tensor           # (C,M,H,W) u8
    n_inst: torch.Tensor          # (C,) i32
    inst_class: torch.Tensor      # (C,M) i32
    inst_box: torch.Tensor        # (C,M,4) f32
    inst_score: torch.Tensor      # (C,M) f32
    boxes3d: torch.Tensor         # (K,8) x y z l w h yaw type (ground truth, unused by the path)


def _boxes(rng: np.random.Generator, k_min=6, k_max=14):
    K = int(rng.integers(k_min, k_max + 1))
    out = []
    tries = 0
    while len(out) < K and tries < 200:
        tries += 1
        t = int(rng.choice(len(BOX_TYPES), p=[0.4, 0.12, 0.2, 0.1, 0.08, 0.1]))
        _, l, w, h = BOX_TYPES[t]
        r = rng.uniform(5.0, 50.0)
        a = rng.uniform(-math.pi, math.pi)
        x, y = r * math.cos(a), r * math.sin(a)
        if any((x - o[0]) ** 2 + (y - o[1]) ** 2 < (0.5 * (l + o[3]) + 0.8) ** 2 for o in out):
            continue
        out.append([x, y, GROUND_Z + h / 2, l, w, h, rng.uniform(-math.pi, math.pi), float(t)])
    return np.array(out, np.float64)


def make_calibration(yaw_deg, H, W, rng: np.random.Generator):
    """Camera with nuScenes-like intrinsics scaled to (H,W) (nuscenes2kitti.py:225-236)."""
    s = W / 1600.0
    fx = 1266.417203046554 * s
    P2 = np.array([[fx, 0, 816.2670197447984 * s, 0.0],
                   [0, fx, 491.50706579294757 * (H / 900.0), 0.0],
                   [0, 0, 1, 0.0]], np.float64)
    yaw = math.radians(yaw_deg)
    Rz = np.array([[math.cos(yaw), math.sin(yaw), 0], [-math.sin(yaw), math.cos(yaw), 0], [0, 0, 1]])
    A = np.array([[0, -1, 0], [0, 0, -1], [1, 0, 0]], np.float64)
    R = A @ Rz
    t = np.array([0.012, -0.33, -0.55]) + rng.normal(0, 0.02, 3)
    V2C = np.hstack([R, t.reshape(3, 1)])
    return Calibration({"P2": P2.astype(np.float32), "R0": np.eye(3, dtype=np.float32),
                        "Tr_velo2cam": V2C.astype(np.float32)})


def _raycast(orig, dirs, boxes, device):
    """orig (3,) or (n,3), dirs (n,3) in the LiDAR frame -> (t (n,), hit id (n,), -1 ground / -2 miss)."""
    n = dirs.shape[0]
    big = 1.0e9
    dz = dirs[:, 2]
    oz = orig[..., 2]
    tg = (GROUND_Z - oz) / torch.where(dz.abs() < 1e-12, torch.full_like(dz, -1e-12), dz)
    t_best = torch.where((tg > 0) & (dz < 0), tg, torch.full_like(tg, big))
    hit = torch.where(t_best < big, torch.full((n,), -1, device=device, dtype=torch.int64),
                      torch.full((n,), -2, device=device, dtype=torch.int64))
    for k in range(boxes.shape[0]):
        x, y, z, l, w, h, yaw = [float(v) for v in boxes[k, :7]]
        c, s = math.cos(yaw), math.sin(yaw)
        o = orig - torch.tensor([x, y, z], device=device, dtype=dirs.dtype)
        ox = o[..., 0] * c + o[..., 1] * s
        oy = -o[..., 0] * s + o[..., 1] * c
        ozl = o[..., 2]
        dx = dirs[:, 0] * c + dirs[:, 1] * s
        dy = -dirs[:, 0] * s + dirs[:, 1] * c
        dzl = dirs[:, 2]
        tmin = torch.full((n,), -big, device=device, dtype=dirs.dtype)
        tmax = torch.full((n,), big, device=device, dtype=dirs.dtype)
        for oo, dd, half in ((ox, dx, l / 2), (oy, dy, w / 2), (ozl, dzl, h / 2)):
            dd = torch.where(dd.abs() < 1e-12, torch.full_like(dd, 1e-12), dd)
            t1 = (-half - oo) / dd
            t2 = (half - oo) / dd
            tmin = torch.maximum(tmin, torch.minimum(t1, t2))
            tmax = torch.minimum(tmax, torch.maximum(t1, t2))
        ok = (tmax >= tmin) & (tmin > 0) & (tmin < t_best)
        t_best = torch.where(ok, tmin, t_best)
        hit = torch.where(ok, torch.full_like(hit, k), hit)
    return t_best, hit


def lidar_sweep(boxes, rng_t: torch.Generator, device, rings=32, az=1085):
    """32 elevation rings in [-30.67, 10.67] deg x `az` azimuth steps (Appendix B)."""
    dt = torch.float64
    el = torch.deg2rad(torch.linspace(-30.67, 10.67, rings, dtype=dt, device=device))
    azs = torch.arange(az, dtype=dt, device=device) * (2 * math.pi / az)
    E, A = torch.meshgrid(el, azs, indexing="ij")
    dirs = torch.stack([torch.cos(E) * torch.cos(A), torch.cos(E) * torch.sin(A), torch.sin(E)], -1).reshape(-1, 3)
    n = dirs.shape[0]
    orig = torch.zeros(3, dtype=dt, device=device)
    t, hit = _raycast(orig, dirs, boxes, device)
    u = torch.rand((n, 3), generator=rng_t, device=device, dtype=dt)
    miss = hit == -2
    t = torch.where(miss | (t > 100.0), 80.0 + 20.0 * u[:, 0], t)       # clutter keeps N constant
    t = t * (1.0 + 0.002 * (u[:, 1] - 0.5))
    pts = dirs * t[:, None]
    gn = hit == -1
    pts[:, 2] = torch.where(gn, pts[:, 2] + 0.04 * (u[:, 2] - 0.5), pts[:, 2])
    inten = torch.rand((n, 1), generator=rng_t, device=device, dtype=dt) * 255.0
    return torch.cat([pts, inten], 1).to(torch.float32)


def camera_view(calib: Calibration, boxes, H, W, M, rng_t: torch.Generator, rng: np.random.Generator,
                device, dense=True, salt=0.005, depth_noise=0.01):
    dt = torch.float64
    vv, uu = torch.meshgrid(torch.arange(H, dtype=dt, device=device),
                            torch.arange(W, dtype=dt, device=device), indexing="ij")
    fu, fv, cu, cv = float(calib.fu), float(calib.fv), float(calib.cu), float(calib.cv)
    # camera (rect) frame rays with z = 1, then into the LiDAR frame
    dr = torch.stack([(uu - cu) / fu, (vv - cv) / fv, torch.ones_like(uu)], -1).reshape(-1, 3)
    R = torch.tensor(calib.V2C[:, :3].astype(np.float64), device=device)     # lidar -> cam
    tvec = torch.tensor(calib.V2C[:, 3].astype(np.float64), device=device)
    dirs = dr @ R                       # R^T applied to row vectors
    orig = -(tvec @ R)
    t, hit = _raycast(orig, dirs, boxes, device)
    depth = torch.where(hit == -2, torch.zeros_like(t), t)            # z_cam = t since ray z = 1
    depth = torch.where(depth > 120.0, torch.zeros_like(depth), depth)
    hit = hit.reshape(H, W)
    # instances: up to M boxes with the largest silhouettes
    K = boxes.shape[0]
    areas = [(int((hit == k).sum().item()), k) for k in range(K)]
    areas = [a for a in areas if a[0] >= 12]
    areas.sort(reverse=True)
    chosen = [k for _, k in areas[:M]]
    order = sorted(chosen)
    masks = torch.zeros((M, H, W), dtype=torch.uint8, device=device)
    inst_class = torch.zeros((M,), dtype=torch.int32)
    inst_box = torch.zeros((M, 4), dtype=torch.float32)
    inst_score = torch.zeros((M,), dtype=torch.float32)
    for j, k in enumerate(order):
        m = hit == k
        # silhouette noise: grow by one pixel in a random direction
        sh = int(rng.integers(0, 4))
        g = torch.roll(m, shifts=(1 if sh == 0 else -1 if sh == 1 else 0,
                                  1 if sh == 2 else -1 if sh == 3 else 0), dims=(0, 1))
        m = m | g
        masks[j] = m.to(torch.uint8)
        ys, xs = torch.nonzero(m, as_tuple=True)
        inst_box[j] = torch.tensor([xs.min().item(), ys.min().item(), xs.max().item(), ys.max().item()],
                                   dtype=torch.float32)
        inst_class[j] = NUSC_CLASSES.index(BOX_TYPES[int(boxes[k, 7])][0])
        inst_score[j] = float(rng.uniform(0.7, 1.0))
    if dense:
        u = torch.rand((H * W, 3), generator=rng_t, device=device, dtype=dt)
        nz = (u[:, 0] + u[:, 1] + u[:, 2] - 1.5) * 2.0            # ~N(0,1), exact arithmetic
        depth = depth * (1.0 + depth_noise * nz)
        sp = torch.rand((H * W,), generator=rng_t, device=device, dtype=dt) < salt
        depth = torch.where(sp & (depth > 0), depth * 0.35, depth)
        depth = depth.reshape(H, W).to(torch.float32)
    else:
        depth = torch.zeros((H, W), dtype=torch.float32, device=device)
    return depth, masks, len(order), inst_class, inst_box, inst_score


def make_scene(seed: int, H=900, W=1600, M=8, cams=6, dense=True, device="cpu", rings=32, az=1085,
               k_min=6, k_max=14) -> Scene:
    rng = np.random.default_rng(1000 + seed)
    device = torch.device(device)
    rng_t = torch.Generator(device=device)
    rng_t.manual_seed(1000 + seed)
    boxes = _boxes(rng, k_min, k_max)
    points = lidar_sweep(boxes, rng_t, device, rings, az)
    calibs, depths, maskl, nl, cl, bl, sl = [], [], [], [], [], [], []
    for c in range(cams):
        cal = make_calibration(CAM_YAWS_DEG[c % len(CAM_YAWS_DEG)], H, W, rng)
        d, m, n, ic, ib, isc = camera_view(cal, boxes, H, W, M, rng_t, rng, device, dense)
        calibs.append(cal); depths.append(d); maskl.append(m); nl.append(n)
        cl.append(ic); bl.append(ib); sl.append(isc)
    return Scene(points, calibs, torch.stack(depths), torch.stack(maskl),
                 torch.tensor(nl, dtype=torch.int32), torch.stack(cl), torch.stack(bl),
                 torch.stack(sl), torch.tensor(boxes))


def to_view_batch(scenes: List[Scene], params: Params, device, dense=True, thing_classes=None, frame_ids=None):
    """Pack scenes (frames) into the engine's ViewBatch on `device`.  frame_ids: the global index of every scene
    (default 0..len-1) -- what the gathered box rows carry."""
    from .engine import ViewBatch
    thing_classes = thing_classes or NUSC_CLASSES
    dev = torch.device(device)
    pts = torch.cat([s.points for s in scenes]).to(dev).contiguous()
    off = np.zeros(len(scenes) + 1, np.int64)
    off[1:] = np.cumsum([s.points.shape[0] for s in scenes])
    vf, recs, cams = [], [], []
    for f, s in enumerate(scenes):
        for c, cal in enumerate(s.calibs):
            vf.append(f)
            cams.append(c)
            recs.append(cal.record())
    V = len(vf)
    M = scenes[0].masks.shape[1]
    inst_class = torch.cat([s.inst_class for s in scenes]).to(torch.int32)
    names = [[thing_classes[int(c)] for c in row] for row in inst_class.tolist()]
    rl = torch.tensor([[params.instance_radii(n)[0] for n in row] for row in names], dtype=torch.float64)
    rp = torch.tensor([[params.instance_radii(n)[1] for n in row] for row in names], dtype=torch.float64)
    is_car = torch.tensor([[1 if n == "Car" else 0 for n in row] for row in names], dtype=torch.int32)
    return ViewBatch(
        points=pts, pt_off=torch.tensor(off, dtype=torch.int32, device=dev),
        view_frame=torch.tensor(vf, dtype=torch.int32, device=dev),
        calib=torch.tensor(np.stack(recs), dtype=torch.float32, device=dev).contiguous(),
        masks=torch.cat([s.masks for s in scenes]).to(dev).contiguous(),
        n_inst=torch.cat([s.n_inst for s in scenes]).to(dev),
        inst_class=inst_class.to(dev).contiguous(), inst_is_car=is_car.to(dev),
        inst_r_lidar=rl.to(dev), inst_r_pseudo=rp.to(dev),
        inst_box=torch.cat([s.inst_box for s in scenes]).to(dev).contiguous(),
        inst_score=torch.cat([s.inst_score for s in scenes]).to(dev).contiguous(),
        view_key=torch.arange(V, dtype=torch.int64, device=dev),
        host_pt_off=off, host_view_frame=np.array(vf, np.int64),
        depth=(torch.cat([s.depth for s in scenes]).to(dev).contiguous() if dense else None),
        frame_ids=np.asarray(frame_ids if frame_ids is not None else range(len(scenes)), np.int64),
        view_cam=np.asarray(cams, np.int64))
