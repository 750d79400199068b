"""CPU oracle for DFU3D's pseudo-box generation path -- TEST INFRASTRUCTURE ONLY.

A NumPy (+ small C helper, oracle/csrc/dfu3d_oracle.c) restatement of the
reference's hot path, stage by stage, dtype-faithful.  Citations are
file:line relative to the reference checkout (tools/PENet/...).

Who may import this: tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- as the checker / reported CPU baseline only.  The product
(dfu3d_amd/) never imports it and fails loudly when its HIP library is missing.

Pinning: golden vectors in tests/golden/*.npz were produced in the build
container by importing the reference's own modules
(tests/golden/capture_goldens.py); tests/test_oracle_golden.py checks this file
against them.  The third-party leaves that are absent from the reference tree
and from this image (Open3D remove_radius_outlier / remove_statistical_outlier,
spconv point-to-voxel) are restated from their published algorithms and are
"parity unpinned" (SURVEY.md §8c); sklearn's RANSAC is replaced by a seeded,
counter-based 3-point RANSAC with the same inlier rule (hazard H1).
"""
from __future__ import annotations

import ctypes
import math
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import build as _build

_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(_build.selected())      # (built if missing or older than its source; DFU3D_ORACLE_SANITIZE=1: the ASan build)
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# --------------------------------------------------------------------------
# canonical parameters (nuScenes variant my_loader.py)
# --------------------------------------------------------------------------
NUSC_CLASSES = ['Car', 'Truck', 'Construction_vehicle', 'Bus', 'Trailer',
                'Barrier', 'Motorcycle', 'Bicycle', 'Pedestrian',
                'Traffic_cone']                      # my_loader.py:634
VEHICLE_CLASSES = ('Car', 'Truck', 'Construction_vehicle', 'Bus', 'Trailer')   # my_loader.py:581
SMALL_CLASSES = ('Pedestrian', 'Bicycle', 'Motorcycle', 'Barrier', 'Traffic_cone')  # my_loader.py:591


@dataclass
class Params:
    depth_min: float = 0.001            # my_loader.py:507
    plane_max_hs: float = 8.0           # my_loader.py:448
    plane_range: float = 54.0           # my_loader.py:448
    plane_offset: float = 0.05          # my_loader.py:471
    bounds_hw: Tuple[int, int] = (900, 1600)   # my_loader.py:526
    fov_hw: Tuple[int, int] = (900, 1600)      # vis_utils.py:153 (H11)
    z_max: float = 1.0                  # my_loader.py:540
    theta_min: float = 1.5              # my_loader.py:175
    # spconv stores voxel size / range as float32 and promotes to double
    vsize: Tuple[float, float, float] = (float(np.float32(200.0)),
                                         float(np.float32(0.002)),
                                         float(np.float32(0.002)))   # my_loader.py:70
    vrange_min: Tuple[float, float, float] = (-100.0, -5.0, -5.0)     # my_loader.py:71
    vgrid: Tuple[int, int, int] = (1, 5000, 5000)
    max_points_per_voxel: int = 100     # my_loader.py:73
    max_voxels: int = 1000000           # my_loader.py:74
    nb_points: int = 1                  # my_loader.py:584
    radius_vehicle: float = 3.0         # my_loader.py:584
    radius_small: float = 0.6           # my_loader.py:593
    small_class_drop_lidar: bool = True  # H4, my_loader.py:592-594
    fuse_C: float = 0.1                 # my_loader.py:603
    R0: float = 3.0                     # rectangle_fitting.py:36
    Rd: float = 0.001                   # rectangle_fitting.py:37
    dtheta_deg: float = 1.0             # rectangle_fitting.py:35
    car_aspect_max: float = 5.0         # my_loader.py:651
    # dormant global statistical filter (my_loader0.py:734-735); off = shipped
    stat_filter: bool = False
    stat_voxel: float = 0.05            # my_loader0.py:734 voxel_down_sample(voxel_size=0.05)
    stat_nb_neighbors: int = 30
    stat_std_ratio: float = 0.3
    # seeded RANSAC (H1)
    ransac_trials: int = 100
    ransac_seed: int = 0


# --------------------------------------------------------------------------
# a3  calibration_kitti.py:23-144
# --------------------------------------------------------------------------
def get_calib_from_file(filepath):
    """calibration_kitti.py:23-58 (float32 parse; default R0 when absent)."""
    R0 = np.array([[0.99992624, 0.00965411, -0.0072371],
                   [-0.00968531, 0.99994343, -0.00433077],
                   [0.00719491, 0.00440054, 0.99996366]])
    P2 = P3 = vtc = None
    with open(filepath) as f:
        for line in f.readlines():
            tok = line.strip().split(" ")
            if line[:2] == "P2":
                P2 = np.array(tok[-12:], np.float32)
            if line[:2] == "P3":
                P3 = np.array(tok[-12:], np.float32)
            if line[:14] == "Tr_velo_to_cam" or line[:11] == "Tr_velo_cam":
                vtc = np.array(tok[-12:], np.float32)
            if line[:7] == "R0_rect" or line[:6] == "R_rect":
                R0 = np.array(tok[-9:], np.float32)
    return {"P2": P2.reshape(3, 4), "P3": P3.reshape(3, 4),
            "Tr_velo2cam": vtc.reshape(3, 4), "R0": R0.reshape(3, 3)}


class Calibration:
    """calibration_kitti.py:62-144, with the BLAS products written out.

    fp32 products are sequential-k FMA chains -- the arithmetic numpy's sgemm
    performs (pinned by tests/test_oracle_golden.py against np.dot).
    """

    def __init__(self, calib):
        if not isinstance(calib, dict):
            calib = get_calib_from_file(calib)
        self.P2 = np.ascontiguousarray(calib['P2'], np.float32)
        self.R0 = np.ascontiguousarray(calib['R0'], np.float32)
        self.V2C = np.ascontiguousarray(calib['Tr_velo2cam'], np.float32)
        self.cu = self.P2[0, 2]
        self.cv = self.P2[1, 2]
        self.fu = self.P2[0, 0]
        self.fv = self.P2[1, 1]
        self.tx = self.P2[0, 3] / (-self.fu)
        self.ty = self.P2[1, 3] / (-self.fv)
        self.M43 = np.empty((4, 3), np.float32)
        _lib().orc_m43_f32(_p(self.V2C), _p(self.R0), _p(self.M43))
        # calibration_kitti.py:95-101: inv((R0_ext @ V2C_ext)^T) in float32
        R0e = np.zeros((4, 4), np.float32)
        R0e[:3, :3] = self.R0
        R0e[3, 3] = 1
        V2Ce = np.zeros((4, 4), np.float32)
        V2Ce[:3, :] = self.V2C
        V2Ce[3, 3] = 1
        self.Minv = np.ascontiguousarray(
            np.linalg.inv(np.dot(R0e, V2Ce).T), np.float32)

    def lidar_to_rect(self, pts_lidar):
        """calibration_kitti.py:104-112."""
        pts_lidar = np.asarray(pts_lidar)
        n = pts_lidar.shape[0]
        if pts_lidar.dtype == np.float32:
            src = np.ascontiguousarray(pts_lidar[:, :3])
            out = np.empty((n, 3), np.float32)
            _lib().orc_lidar_to_rect_f32(_p(src), ctypes.c_int64(3),
                                         _p(self.M43), _p(out),
                                         ctypes.c_int64(n))
            return out
        return self._hom_dot_f64(pts_lidar, self.M43, 3)

    @staticmethod
    def _hom_dot_f64(pts, M, m):
        """[xyz,1] float64 @ float32 (4,ld) matrix, first m columns: the sequential-k
        FMA chain numpy's dgemm performs (pinned bit-exactly by golden G1)."""
        p = np.ascontiguousarray(np.asarray(pts, np.float64)[:, :3])
        M = np.ascontiguousarray(M, np.float32)
        out = np.empty((p.shape[0], m), np.float64)
        _lib().orc_hom_dot_f64(_p(p), _p(M), ctypes.c_int64(M.shape[1]), ctypes.c_int64(m), _p(out),
                               ctypes.c_int64(p.shape[0]))
        return out

    def rect_to_img(self, pts_rect):
        """calibration_kitti.py:114-123 (float32 path)."""
        pts_rect = np.ascontiguousarray(pts_rect, np.float32)
        n = pts_rect.shape[0]
        uv = np.empty((n, 2), np.float32)
        d = np.empty((n,), np.float32)
        _lib().orc_rect_to_img_f32(_p(pts_rect), _p(self.P2), _p(uv), _p(d),
                                   ctypes.c_int64(n))
        return uv, d

    def lidar_to_img(self, pts_lidar):
        """calibration_kitti.py:125-132."""
        return self.rect_to_img(self.lidar_to_rect(pts_lidar))

    def img_to_rect(self, u, v, depth_rect):
        """calibration_kitti.py:134-144 (int64 pixels, f32 scalars -> fp64)."""
        x = ((u - self.cu) * depth_rect) / self.fu + self.tx
        y = ((v - self.cv) * depth_rect) / self.fv + self.ty
        return np.concatenate((x.reshape(-1, 1), y.reshape(-1, 1),
                               depth_rect.reshape(-1, 1)), axis=1)

    def rect_to_lidar(self, pts_rect):
        """calibration_kitti.py:89-102 (fp64 points x float32 inverse; dgemm == FMA chain)."""
        return self._hom_dot_f64(pts_rect, self.Minv, 3)


# --------------------------------------------------------------------------
# a4  vis_utils.py:108-123 / my_loader.py:85-100
# --------------------------------------------------------------------------
def get_fov_flag(pts_rect, img_shape, calib):
    pts_img, depth = calib.rect_to_img(pts_rect)
    f1 = np.logical_and(pts_img[:, 0] >= 0, pts_img[:, 0] < img_shape[1])
    f2 = np.logical_and(pts_img[:, 1] >= 0, pts_img[:, 1] < img_shape[0])
    return np.logical_and(np.logical_and(f1, f2), depth >= 0)


# --------------------------------------------------------------------------
# a5  my_loader.py:448-484 (plane); RANSAC per hazard H1
# --------------------------------------------------------------------------
_M64 = (1 << 64) - 1


def _mix64(z):
    z = (z + 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def ransac_draw(seed, key, trial, attempt, n):
    """Counter-based index in [0, n): mulhi(mix(...), n)."""
    r = _mix64((seed ^ _mix64((key << 20) ^ (trial << 8) ^ attempt)) & _M64)
    return (r * n) >> 64


def plane_ransac(xyz, params: Params = None, key: int = 0):
    """Seeded replacement for estimate_plane (my_loader.py:448-469).

    Same candidate set (z < max_hs, |x|,|y| < 54; :449-453), same inlier rule as
    sklearn's RANSACRegressor defaults (|residual| <= MAD(z), 3-point samples,
    most inliers wins, final least-squares refit on the inliers), but a fixed
    number of trials drawn from a counter-based generator so that the HIP
    kernel and this oracle take the same samples.  Returns the (4,) plane
    (unit normal pointing up, offset) exactly as :457-466 builds it.
    """
    params = params or Params()
    xyz = np.asarray(xyz)
    m = ((xyz[:, 2] < params.plane_max_hs)
         & (xyz[:, 0] > -params.plane_range) & (xyz[:, 0] < params.plane_range)
         & (xyz[:, 1] > -params.plane_range) & (xyz[:, 1] < params.plane_range))
    p = xyz[m].astype(np.float64)
    n = p.shape[0]
    if n < 3:
        return np.array([0.0, 0.0, 1.0, 1e30])
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    thr = float(np.median(np.abs(z - np.median(z))))
    best_cnt, best = -1, None
    for t in range(params.ransac_trials):
        idx, a = [], 0
        while len(idx) < 3 and a < 64:
            i = ransac_draw(params.ransac_seed, key, t, a, n)
            a += 1
            if i not in idx:
                idx.append(i)
        if len(idx) < 3:
            continue
        i0, i1, i2 = idx
        dx1, dy1, dz1 = x[i1] - x[i0], y[i1] - y[i0], z[i1] - z[i0]
        dx2, dy2, dz2 = x[i2] - x[i0], y[i2] - y[i0], z[i2] - z[i0]
        det = dx1 * dy2 - dx2 * dy1
        if not abs(det) > 1e-12:
            continue
        ca = (dz1 * dy2 - dz2 * dy1) / det
        cb = (dx1 * dz2 - dx2 * dz1) / det
        cc = (z[i0] - ca * x[i0]) - cb * y[i0]
        res = np.abs(z - ((ca * x + cb * y) + cc))
        cnt = int((res <= thr).sum())
        if cnt > best_cnt:
            best_cnt, best = cnt, (ca, cb, cc)
    if best is None:
        return np.array([0.0, 0.0, 1.0, 1e30])
    ca, cb, cc = best
    inl = np.abs(z - ((ca * x + cb * y) + cc)) <= thr
    xi, yi, zi = x[inl], y[inl], z[inl]
    k = xi.shape[0]
    mx, my, mz = xi.sum() / k, yi.sum() / k, zi.sum() / k
    ux, uy, uz = xi - mx, yi - my, zi - mz
    sxx, sxy, syy = (ux * ux).sum(), (ux * uy).sum(), (uy * uy).sum()
    sxz, syz = (ux * uz).sum(), (uy * uz).sum()
    det2 = sxx * syy - sxy * sxy
    if det2 > 1e-12 * max(sxx * syy, 1e-300):
        ca = (sxz * syy - syz * sxy) / det2
        cb = (syz * sxx - sxz * sxy) / det2
        cc = (mz - ca * mx) - cb * my
    # my_loader.py:457-466
    norm = math.sqrt((ca * ca + cb * cb) + 1.0)
    return np.array([-(ca / norm), -(cb / norm), 1.0 / norm, -(cc / norm)])


def distance_to_plane(ptc, plane, directional=False):
    """my_loader.py:479-484."""
    ptc = np.asarray(ptc, np.float64)
    d = ((ptc[:, 0] * plane[0] + ptc[:, 1] * plane[1]) + ptc[:, 2] * plane[2]) + plane[3]
    if not directional:
        d = np.abs(d)
    d = d / np.sqrt((plane[0] * plane[0] + plane[1] * plane[1]) + plane[2] * plane[2])
    return d


def above_plane(ptc, plane, offset=0.05, only_range=54.0):
    """my_loader.py:471-477."""
    mask = distance_to_plane(ptc, plane, directional=True) < offset
    if only_range is not None:
        rm = ((ptc[:, 0] < only_range) & (ptc[:, 0] > -only_range)
              & (ptc[:, 1] < only_range) & (ptc[:, 1] > -only_range))
        mask = mask & rm
    return np.logical_not(mask)


# --------------------------------------------------------------------------
# a6  my_loader.py:517-530 point->pixel label inheritance (hazard H3)
# --------------------------------------------------------------------------
def label_inheritance(lidar_mask, calib, masks_u8, bounds_hw=(900, 1600)):
    """Returns (list of per-instance row-index arrays into lidar_mask, K).

    uv = round_half_even(lidar_to_img); the values of uint8(mask_j) at the
    in-bounds pixels are COMPACTED and applied positionally to rows 0..K-1
    (my_loader.py:526-530); lidar_mask is truncated to K rows for all j.
    """
    H, W = bounds_hw
    pts_img, _ = calib.lidar_to_img(lidar_mask[:, :3])
    r = np.round(pts_img)
    inb = (0 <= r[:, 0]) & (r[:, 0] < W) & (0 <= r[:, 1]) & (r[:, 1] < H)
    pu = r[inb, 0].astype(np.int64)
    pv = r[inb, 1].astype(np.int64)
    K = min(int(inb.sum()), lidar_mask.shape[0])
    out = []
    for j in range(masks_u8.shape[0]):
        vals = masks_u8[j][pv[:K], pu[:K]]
        out.append(np.nonzero(vals > 0)[0])
    return out, K


# --------------------------------------------------------------------------
# a7  my_loader.py:507-509, 532-540 pixel -> 3-D back-projection
# --------------------------------------------------------------------------
def backproject(depth_hw, calib, depth_min=0.001):
    """Returns (rows, cols, p_lidar fp64 (P0,3)); mutates depth like :507."""
    depth_hw[depth_hw < depth_min] = 0
    rows, cols = np.nonzero(depth_hw)
    depth_val = depth_hw[rows, cols]
    p_rect = calib.img_to_rect(cols, rows, depth_val)
    return rows, cols, calib.rect_to_lidar(p_rect)


# --------------------------------------------------------------------------
# a8  my_loader.py:166-180, 247-275, 22-83 spherical voxel sampling
# --------------------------------------------------------------------------
def to_sphere_coords(xyz):
    """my_loader.py:166-180 -> (r, theta, phi, keep mask theta > 1.5)."""
    with np.errstate(all='ignore'):
        s = xyz[:, 0] * xyz[:, 0]
        s = s + xyz[:, 1] * xyz[:, 1]
        s = s + xyz[:, 2] * xyz[:, 2]
        r = np.sqrt(s)
        theta = np.arccos(xyz[:, 2] / r)
        phi = np.arctan(xyz[:, 1] / xyz[:, 0])
    return r, theta, phi


_VOX_TABLE = {}


def voxel_sample(xyz, key, params: Params = None):
    """la_sampling2 / la_sampling20 selection (my_loader.py:247-275).

    xyz: (n,3) fp64 points in input order; key: (n,) fp64 column whose argmin
    picks the representative (z for la_sampling2, y for la_sampling20; H8).
    Returns indices (into xyz) of the representatives in first-seen voxel
    order.
    """
    params = params or Params()
    n = xyz.shape[0]
    if n == 0:
        return np.zeros((0,), np.int64)
    r, theta, phi = to_sphere_coords(xyz)
    keep = theta > params.theta_min           # my_loader.py:175
    sel = np.nonzero(keep)[0]
    coords = np.ascontiguousarray(
        np.stack([r[sel], theta[sel], phi[sel]], axis=1), np.float64)
    m = coords.shape[0]
    grid = np.array(params.vgrid, np.int32)
    gsz = int(grid[0]) * int(grid[1]) * int(grid[2])
    tab = _VOX_TABLE.get(gsz)
    if tab is None:
        tab = np.full((gsz,), -1, np.int32)
        _VOX_TABLE[gsz] = tab
    vox_of_pt = np.empty((m,), np.int32)
    cap = min(m, params.max_voxels)
    vox_count = np.empty((max(cap, 1),), np.int32)
    vox_cell = np.empty((max(cap, 1),), np.int64)
    vs = np.array(params.vsize, np.float64)
    rmin = np.array(params.vrange_min, np.float64)
    nv = _lib().orc_voxelize(_p(coords), ctypes.c_int64(3), ctypes.c_int64(m),
                             _p(vs), _p(rmin), _p(grid),
                             ctypes.c_int32(params.max_points_per_voxel),
                             ctypes.c_int32(params.max_voxels), _p(tab),
                             _p(vox_of_pt), _p(vox_count), _p(vox_cell))
    rep = np.empty((max(nv, 1),), np.int64)
    k = np.ascontiguousarray(key[sel], np.float64)
    _lib().orc_voxel_argmin(_p(vox_of_pt), _p(k), ctypes.c_int64(m),
                            ctypes.c_int64(nv), _p(rep))
    return sel[rep[:nv]]


# --------------------------------------------------------------------------
# a10 / a11  Open3D outlier filters (third-party leaves)
# --------------------------------------------------------------------------
def radius_outlier(pts, nb_points, radius):
    """remove_radius_outlier: returns kept indices (order preserved)."""
    pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 3)
    n = pts.shape[0]
    keep = np.zeros((n,), np.uint8)
    if n:
        _lib().orc_radius_outlier(_p(pts), ctypes.c_int64(n),
                                  ctypes.c_double(radius),
                                  ctypes.c_int32(nb_points), _p(keep))
    return np.nonzero(keep)[0]


def voxel_down_sample(pts, voxel_size):
    """Open3D PointCloud::VoxelDownSample(voxel_size) (my_loader0.py:734, commented; the leaf is absent from the reference
    tree and from this image -- restated from Open3D's published source, geometry/PointCloud.cpp, PARITY UNPINNED):
        voxel_min_bound = GetMinBound() - voxel_size * 0.5
        voxel_index     = int(floor((p - voxel_min_bound) / voxel_size))          per axis
        voxelindex_to_accpoint[voxel_index].AddPoint(p)   for the points in input order (point_ += p; num_of_points_++)
        output          = point_ / double(num_of_points_)                         per occupied voxel
    Open3D emits the voxels in the iteration order of an std::unordered_map, which is unspecified.  DEFINED here (and in
    the library, include/dfu3d.h): first-seen order -- a voxel's place is that of its first point in the input list.
    Returns the (m,3) centroids."""
    pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 3)
    if pts.shape[0] == 0:
        return pts
    if not voxel_size > 0.0:
        raise ValueError("voxel_size <= 0")                       # Open3D: LogError
    minb = pts.min(0) - voxel_size * 0.5
    idx = np.floor((pts - minb) / voxel_size).astype(np.int64)
    _, first, inv = np.unique(idx, axis=0, return_index=True, return_inverse=True)
    inv = inv.reshape(-1)
    order = np.argsort(first, kind="stable")                      # voxels by the position of their first point
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    slot = rank[inv]
    acc = np.zeros((order.size, 3), np.float64)
    np.add.at(acc, slot, pts)                                     # unbuffered: acc[slot[i]] += pts[i] for i = 0, 1, 2, ... in input order
    cnt = np.bincount(slot, minlength=order.size).astype(np.float64)
    return acc / cnt[:, None]


def statistical_outlier(pts, nb_neighbors, std_ratio):
    """remove_statistical_outlier: returns kept indices (order preserved)."""
    pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 3)
    n = pts.shape[0]
    if n == 0 or nb_neighbors < 1:
        return np.zeros((0,), np.int64)
    md = np.empty((n,), np.float64)
    _lib().orc_knn_mean_dist(_p(pts), ctypes.c_int64(n),
                             ctypes.c_int32(nb_neighbors), _p(md))
    valid = n
    cloud_mean = 0.0
    for v in md:                       # std::accumulate order
        cloud_mean += v if v > 0 else 0.0
    cloud_mean /= valid
    sq = 0.0
    for v in md:
        sq += (v - cloud_mean) * (v - cloud_mean) if v > 0 else 0.0
    std = math.sqrt(sq / (valid - 1)) if valid > 1 else float('nan')
    thr = cloud_mean + std_ratio * std
    return np.nonzero((md > 0) & (md < thr))[0]


# --------------------------------------------------------------------------
# a12  my_loader.py:489-494 BallQuery (roi_max_dim == 0; strict <)
# --------------------------------------------------------------------------
def ball_query(p1, p2, C):
    p1 = np.ascontiguousarray(p1, np.float64).reshape(-1, 3)
    p2 = np.ascontiguousarray(p2, np.float64).reshape(-1, 3)
    keep = np.zeros((p1.shape[0],), np.uint8)
    if p1.shape[0]:
        _lib().orc_ball_query(_p(p1), ctypes.c_int64(p1.shape[0]), _p(p2),
                              ctypes.c_int64(p2.shape[0]), ctypes.c_double(C),
                              _p(keep))
    return keep.astype(bool)


# --------------------------------------------------------------------------
# a13  rectangle_fitting.py:161-191
# --------------------------------------------------------------------------
def range_cluster_labels(x, y, R0=3.0, Rd=0.001):
    """label[i] = smallest index in i's cluster."""
    x = np.ascontiguousarray(x, np.float64)
    y = np.ascontiguousarray(y, np.float64)
    n = x.shape[0]
    lab = np.empty((n,), np.int32)
    if n:
        _lib().orc_range_cluster(_p(x), _p(y), ctypes.c_int64(n),
                                 ctypes.c_double(R0), ctypes.c_double(Rd),
                                 _p(lab))
    return lab


def range_segmentation(x, y, R0=3.0, Rd=0.001):
    """-> list of index arrays, ordered by ascending smallest index
    (the order rectangle_fitting.py:179-183's merge loop leaves behind)."""
    lab = range_cluster_labels(x, y, R0, Rd)
    roots = np.unique(lab)
    return [np.nonzero(lab == r)[0] for r in roots]


# --------------------------------------------------------------------------
# a14  rectangle_fitting.py:83-159
# --------------------------------------------------------------------------
def variance_criterion(c1, c2):
    """rectangle_fitting.py:83-111."""
    c1max, c1min, c2max, c2min = c1.max(), c1.min(), c2.max(), c2.min()
    D1 = np.minimum(np.abs(c1max - c1), np.abs(c1 - c1min))
    D2 = np.minimum(np.abs(c2max - c2), np.abs(c2 - c2min))
    sel = D1 < D2
    E1, E2 = D1[sel], D2[~sel]
    V1 = -np.var(E1) if E1.size else 0.0
    V2 = -np.var(E2) if E2.size else 0.0
    return V1 + V2


def rectangle_search(x, y, dtheta_deg=1.0):
    """rectangle_fitting.py:113-159 -> (theta*, a[4], b[4], c[4])."""
    x = np.asarray(x, np.float64)
    y = np.asarray(y, np.float64)
    dtheta = np.deg2rad(dtheta_deg)
    best = (-float('inf'), None)
    for theta in np.arange(0.0, np.pi / 2.0 - dtheta, dtheta):
        c, s = np.cos(theta), np.sin(theta)
        c1 = x * c + y * s
        c2 = x * (-s) + y * c
        cost = variance_criterion(c1, c2)
        if best[0] < cost:
            best = (cost, theta)
    th = best[1]
    s, c = np.sin(th), np.cos(th)
    c1 = x * c + y * s
    c2 = x * (-s) + y * c
    a = [c, -s, c, -s]
    b = [s, c, s, c]
    cc = [c1.min(), c2.min(), c1.max(), c2.max()]
    return th, a, b, cc


def calc_cross_point(a, b, c):
    """my_loader.py:699-702."""
    x = (b[0] * -c[1] - b[1] * -c[0]) / (a[0] * b[1] - a[1] * b[0])
    y = (a[1] * -c[0] - a[0] * -c[1]) / (a[0] * b[1] - a[1] * b[0])
    return x, y


def calc_rect_contour(a, b, c):
    """my_loader.py:686-697."""
    xs, ys = [None] * 4, [None] * 4
    xs[0], ys[0] = calc_cross_point(a[0:2], b[0:2], c[0:2])
    xs[1], ys[1] = calc_cross_point(a[1:3], b[1:3], c[1:3])
    xs[2], ys[2] = calc_cross_point(a[2:4], b[2:4], c[2:4])
    xs[3], ys[3] = calc_cross_point([a[3], a[0]], [b[3], b[0]], [c[3], c[0]])
    return xs, ys


# --------------------------------------------------------------------------
# a15  my_loader.py:633-684 GenerateAnns
# --------------------------------------------------------------------------
@dataclass
class BoxRow:
    """One KITTI label row (my_loader.py:680) plus provenance."""
    name: str
    alpha: float
    bbox: Tuple[float, float, float, float]
    h: float
    w: float
    l: float
    x: float
    y: float
    z: float
    ry: float
    inst: int = -1
    cluster: int = -1
    cls: int = -1

    def as_vector(self):
        return np.array([self.alpha, *self.bbox, self.h, self.w, self.l,
                         self.x, self.y, self.z, self.ry], np.float64)


def generate_anns(seem_class_name, pts, class_idx, box2d, calib,
                  params: Params = None, inst=-1) -> List[BoxRow]:
    params = params or Params()
    rows = []
    if pts.shape[0] == 0:
        return rows
    x = np.asarray(pts[:, 0], np.float64)
    y = np.asarray(pts[:, 1], np.float64)
    zmax = float(np.asarray(pts[:, 2], np.float64).max())
    clusters = range_segmentation(x, y, params.R0, params.Rd)
    with np.errstate(all='ignore'):
        for ci, ids in enumerate(clusters):
            _, a, b, c = rectangle_search(x[ids], y[ids], params.dtheta_deg)
            cx, cy = calc_rect_contour(a, b, c)
            center_x = (cx[0] + cx[2]) / 2
            center_y = (cy[0] + cy[2]) / 2
            center_z = zmax / 2 - 1.5                     # my_loader.py:647
            height = zmax                                 # my_loader.py:648
            l1 = np.sqrt((cx[0] - cx[3]) ** 2 + (cy[0] - cy[3]) ** 2)
            l2 = np.sqrt((cx[0] - cx[1]) ** 2 + (cy[0] - cy[1]) ** 2)
            if seem_class_name == "Car" and (l1 / l2 > params.car_aspect_max
                                             or l2 / l1 > params.car_aspect_max):
                continue
            if l1 >= l2:
                length, width = l1, l2
                rotation = np.arctan((cy[3] - cy[0]) / (cx[3] - cx[0] + 1e-8))
            elif l1 < l2:
                length, width = l2, l1
                rotation = np.arctan((cy[1] - cy[0]) / (cx[1] - cx[0] + 1e-8))
            else:       # NaN extents: the reference raises UnboundLocalError
                continue
            rotation = -rotation - np.pi / 2
            theta = np.arctan((-center_x / (center_y + 1e-8)))
            alpha = rotation - theta
            ctr = calib.lidar_to_rect(np.array([[center_x, center_y, center_z]],
                                               np.float64))[0]
            rows.append(BoxRow(NUSC_CLASSES[class_idx], float(alpha),
                               tuple(float(v) for v in box2d), float(height),
                               float(width), float(length), float(ctr[0]),
                               float(ctr[1]), float(ctr[2]), float(rotation),
                               inst=inst, cluster=ci, cls=int(class_idx)))
    return rows


# --------------------------------------------------------------------------
# my_loader.py:502-617 depth2pointsrgbpm (orchestration)
# --------------------------------------------------------------------------
@dataclass
class FrameResult:
    rows: List[BoxRow]
    all_points: Optional[np.ndarray]
    plane: np.ndarray
    dbg: dict = field(default_factory=dict)


def instance_points(seem_name, lidar_pts, pseudo_pts, params: Params):
    """my_loader.py:579-605 ('multi-modal' branch): filters + fuse + concat."""
    lidar_pts = np.asarray(lidar_pts, np.float64).reshape(-1, 3)
    pseudo_pts = np.asarray(pseudo_pts, np.float64).reshape(-1, 3)
    if seem_name in VEHICLE_CLASSES:
        L = lidar_pts[radius_outlier(lidar_pts, params.nb_points, params.radius_vehicle)]
        P = pseudo_pts[radius_outlier(pseudo_pts, params.nb_points, params.radius_vehicle)]
    elif seem_name in SMALL_CLASSES:
        if params.small_class_drop_lidar:      # H4: empty pcd at :592-594
            L = np.zeros((0, 3))
        else:
            L = lidar_pts[radius_outlier(lidar_pts, params.nb_points, params.radius_small)]
        P = pseudo_pts[radius_outlier(pseudo_pts, params.nb_points, params.radius_small)]
    else:                                       # H5: undefined in the reference
        L, P = lidar_pts, pseudo_pts
    if params.stat_filter:                      # the reference's dormant pair, my_loader0.py:734-735
        P = voxel_down_sample(P, params.stat_voxel)
        P = P[statistical_outlier(P, params.stat_nb_neighbors, params.stat_std_ratio)]
    if L.shape[0] > 0 and P.shape[0] > 0:       # my_loader.py:602-604
        P = P[ball_query(P, L, params.fuse_C)]
    return np.concatenate([L, P], 0), L, P


def depth2pointsrgbpm(depth, image, calib, lidar, thing_classes, masks,
                      classes, boxes2d, params: Params = None, plane=None,
                      plane_key: int = 0, want_points: bool = True,
                      debug: bool = False) -> FrameResult:
    """Restatement of my_loader.py:502-617 for ONE camera view.

    depth (H,W,1) or (H,W) f32 (mutated like :507); image (H,W,3) int or None;
    lidar (n,4) f32 already FOV-filtered (vis_utils.py:152-154); masks
    (M,H,W) {0,1}; classes (M,) indices into thing_classes; boxes2d (M,4).
    """
    params = params or Params()
    depth_hw = depth.reshape(depth.shape[0], depth.shape[1])
    masks = np.asarray(masks)
    M = masks.shape[0]
    masks_u8 = masks.astype(np.uint8)               # my_loader.py:525
    if plane is None:
        plane = plane_ransac(lidar[:, :3], params, plane_key)   # :511
    pm = above_plane(lidar[:, :3], plane, params.plane_offset, params.plane_range)
    lidar_mask = lidar[pm]                          # :513
    inst_rows, K = label_inheritance(lidar_mask, calib, masks_u8, params.bounds_hw)
    lidar_mask = lidar_mask[:K]

    rows_px, cols_px, p_lidar = backproject(depth_hw, calib, params.depth_min)
    zkeep = p_lidar[:, 2] < params.z_max            # :540
    p0 = p_lidar[zkeep]
    r0, c0 = rows_px[zkeep], cols_px[zkeep]
    rep_y = voxel_sample(p0, p0[:, 1], params)      # la_sampling20 (:556), H19
    rows: List[BoxRow] = []
    dbg = {}
    if debug:
        dbg.update(K=K, n_ag=int(pm.sum()), inst_lidar=[], inst_pseudo=[],
                   inst_points=[], rep_pix=(r0[rep_y] * depth_hw.shape[1] + c0[rep_y]))
    for i in range(M):
        name = thing_classes[int(classes[i])]       # :551
        sel = masks[i][r0[rep_y], c0[rep_y]] > 0.0  # :554-557
        pseudo_i = p0[rep_y[sel]]
        lidar_i = lidar_mask[inst_rows[i]][:, :3]
        pts, L, P = instance_points(name, lidar_i, pseudo_i, params)
        if debug:
            dbg['inst_lidar'].append(inst_rows[i])
            dbg['inst_pseudo'].append(pseudo_i)
            dbg['inst_points'].append(pts)
        rows += generate_anns(name, pts, int(classes[i]), boxes2d[i], calib,
                              params, inst=i)
    all_points = None
    if want_points:
        rep_z = voxel_sample(p0, p0[:, 2], params)  # la_sampling2 (:541)
        new_p1 = np.zeros((rep_z.shape[0], 8))
        new_p1[:, 0:3] = p0[rep_z]
        if image is not None:
            new_p1[:, 4:7] = image[r0[rep_z], c0[rep_z]] / 3   # :539
        new_p1[:, -1] = 1                            # :542
        new_lidar = np.zeros((lidar.shape[0], 8))   # :611-615
        new_lidar[:, 0:4] = lidar[:, 0:4]
        new_lidar[:, 3] *= 10
        new_lidar[:, -1] = 2
        all_points = np.concatenate([new_lidar, new_p1], 0)
    return FrameResult(rows, all_points, np.asarray(plane, np.float64), dbg)


def fov_filter(lidar, calib, fov_hw):
    """vis_utils.py:152-154."""
    flag = get_fov_flag(calib.lidar_to_rect(lidar[:, 0:3]), fov_hw, calib)
    return lidar[flag], flag


def format_row(r: BoxRow) -> str:
    """my_loader.py:680 (15 fields, Python repr floats; hazard H17)."""
    f = [r.name, '0', '0', repr(float(r.alpha)),
         *[repr(float(np.float32(v))) for v in r.bbox],
         repr(float(r.h)), repr(float(r.w)), repr(float(r.l)),
         repr(float(r.x)), repr(float(r.y)), repr(float(r.z)),
         repr(float(r.ry))]
    return ' '.join(f)
