"""Host mirror of the reference's calibration object
(tools/PENet/dataloaders/calibration_kitti.py:23-144): same constructor, same
attribute names, same methods with the reference's dtype behaviour.  On the hot
path the transforms run on the GPU (csrc/common.hpp) from the 48-float per-view
record this class packs (include/dfu3d.h); the host methods below serve the
reference's own callers (vis_utils.py:152-154, my_loader.py:518,535-536,679) so
that the module swap of INTEGRATION.md works unchanged.

float32 products are evaluated as the sequential-k fused-multiply-add chain that
the reference's np.dot (BLAS sgemm) performs for these shapes (pinned bit-exactly
by golden G1), written out so that the result does not depend on the host's BLAS.
"""
from fractions import Fraction

import numpy as np


def get_calib_from_file(filepath):
    """calibration_kitti.py:23-58: float32 parse of P2 / P3 / R0_rect /
    Tr_velo_to_cam; the reference's default R0 when the file has none."""
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
    if P2 is None or vtc is None:
        raise ValueError("%s: P2 / Tr_velo_to_cam missing" % filepath)
    if P3 is None:
        P3 = P2
    return {"P2": P2.reshape(3, 4), "P3": P3.reshape(3, 4),
            "Tr_velo2cam": vtc.reshape(3, 4), "R0": R0.reshape(3, 3)}


def _round_f32(q: Fraction) -> np.float32:
    """Correctly rounded (nearest-even) float32 of an exact rational."""
    if q == 0:
        return np.float32(0.0)
    c = np.float32(float(q))            # within one float32 ulp of q
    best = c
    for cand in (np.nextafter(c, np.float32(-np.inf)), np.nextafter(c, np.float32(np.inf))):
        if not np.isfinite(cand):
            continue
        db, dc = abs(Fraction(float(best)) - q), abs(Fraction(float(cand)) - q)
        if dc < db or (dc == db and (int(np.float32(cand).view(np.uint32)) & 1) == 0
                       and (int(np.float32(best).view(np.uint32)) & 1) == 1):
            best = cand
    return np.float32(best)


def fma_chain_matmul_f32(A, B):
    """(m,k)@(k,n) float32 with a sequential-k fused-multiply-add chain per
    output -- the arithmetic of the reference's np.dot (BLAS sgemm), written
    out so the result does not depend on the host's BLAS."""
    A = np.asarray(A, np.float32)
    B = np.asarray(B, np.float32)
    m, k = A.shape
    n = B.shape[1]
    out = np.zeros((m, n), np.float32)
    for i in range(m):
        for j in range(n):
            acc = np.float32(A[i, 0] * B[0, j])
            for t in range(1, k):
                acc = _round_f32(Fraction(float(A[i, t])) * Fraction(float(B[t, j]))
                                 + Fraction(float(acc)))
            out[i, j] = acc
    return out


def fma_f32(a, b, c):
    """Correctly rounded float32 fma(a, b, c), vectorised.  a*b is exact in
    float64; the float64 sum is turned into its round-to-odd form (the TwoSum
    residual acts as the sticky bit), so the final conversion to float32 is the
    single rounding of the exact a*b+c (53 >= 24 + 2 bits)."""
    p = np.asarray(a, np.float32).astype(np.float64) * np.asarray(b, np.float32).astype(np.float64)
    c = np.asarray(c, np.float32).astype(np.float64)
    with np.errstate(all="ignore"):
        s = p + c
        bb = s - p
        e = (p - (s - bb)) + (c - bb)                 # s + e == p + c exactly
        s = np.atleast_1d(s)
        e = np.broadcast_to(e, s.shape)
        other = np.where(e > 0, np.nextafter(s, np.inf), np.nextafter(s, -np.inf))
        even = (s.view(np.int64) & 1) == 0
        t = np.where((e != 0) & even & np.isfinite(s), other, s)
        return t.astype(np.float32)


def chain_matmul_f32(A, B):
    """(n,k) @ (k,m) float32 as the sequential-k FMA chain per output element
    (== sgemm for the small k of this module), vectorised over n."""
    A = np.asarray(A, np.float32)
    B = np.asarray(B, np.float32)
    n, k = A.shape
    out = np.empty((n, B.shape[1]), np.float32)
    for j in range(B.shape[1]):
        acc = A[:, 0] * B[0, j]
        for t in range(1, k):
            acc = fma_f32(A[:, t], np.broadcast_to(B[t, j], (n,)), acc)
        out[:, j] = acc
    return out


class Calibration(object):
    """calibration_kitti.py:62-144 + the packed GPU record."""

    def __init__(self, calib_file):
        calib = calib_file if isinstance(calib_file, dict) else get_calib_from_file(calib_file)
        self.P2 = np.ascontiguousarray(calib['P2'], np.float32)      # 3 x 4
        self.R0 = np.ascontiguousarray(calib['R0'], np.float32)      # 3 x 3
        self.V2C = np.ascontiguousarray(calib['Tr_velo2cam'], np.float32)  # 3 x 4
        self.cu = self.P2[0, 2]
        self.cv = self.P2[1, 2]
        self.fu = self.P2[0, 0]
        self.fv = self.P2[1, 1]
        self.tx = self.P2[0, 3] / (-self.fu)
        self.ty = self.P2[1, 3] / (-self.fv)
        # calibration_kitti.py:110: np.dot(V2C.T, R0.T), float32
        self.M43 = fma_chain_matmul_f32(self.V2C.T, self.R0.T)
        # calibration_kitti.py:95-101: inv((R0_ext @ V2C_ext)^T), float32
        R0e = np.zeros((4, 4), np.float32)
        R0e[:3, :3] = self.R0
        R0e[3, 3] = 1
        V2Ce = np.zeros((4, 4), np.float32)
        V2Ce[:3, :] = self.V2C
        V2Ce[3, 3] = 1
        self.Minv = np.ascontiguousarray(np.linalg.inv(np.dot(R0e, V2Ce).T), np.float32)

    # ---- the reference's methods (calibration_kitti.py:81-164) -----------------
    def cart_to_hom(self, pts):
        """(N, 3 or 2) -> (N, 4 or 3): a float32 column of ones is appended (:81-87),
        so float32 points stay float32 and float64 points stay float64."""
        pts = np.asarray(pts)
        return np.hstack((pts, np.ones((pts.shape[0], 1), dtype=np.float32)))

    @staticmethod
    def _dot(hom, M):
        """hom @ M with M float32: float32 chain for float32 points, float64 otherwise."""
        if hom.dtype == np.float32:
            return chain_matmul_f32(hom, M)
        return np.dot(hom.astype(np.float64, copy=False), np.asarray(M, np.float64))

    def lidar_to_rect(self, pts_lidar):
        """(N,3) LiDAR -> (N,3) rectified camera: [xyz,1] @ (V2C^T @ R0^T)  (:104-112)."""
        return self._dot(self.cart_to_hom(np.asarray(pts_lidar)[:, 0:3]), self.M43)

    def rect_to_img(self, pts_rect):
        """(N,3) rect -> ((N,2) pixel u,v, (N,) depth): [rect,1] @ P2^T, divided by
        rect z; depth = z' - P2[2,3]  (:114-123)."""
        hom = self.cart_to_hom(pts_rect)
        h = self._dot(hom, self.P2.T)
        with np.errstate(all="ignore"):
            img = (h[:, 0:2].T / hom[:, 2]).T
        return img, h[:, 2] - self.P2.T[3, 2]

    def lidar_to_img(self, pts_lidar):
        """(:125-132)"""
        return self.rect_to_img(self.lidar_to_rect(pts_lidar))

    def img_to_rect(self, u, v, depth_rect):
        """pixel (u, v) + depth -> (N,3) rect (:134-144); integer pixels with the
        float32 intrinsics give float64, as in the reference."""
        x = ((u - self.cu) * depth_rect) / self.fu + self.tx
        y = ((v - self.cv) * depth_rect) / self.fv + self.ty
        return np.concatenate((x.reshape(-1, 1), y.reshape(-1, 1), depth_rect.reshape(-1, 1)), axis=1)

    def rect_to_lidar(self, pts_rect):
        """(N,3) rect -> (N,3) LiDAR: [rect,1] @ inv((R0_ext @ V2C_ext)^T)  (:89-102)."""
        return self._dot(self.cart_to_hom(pts_rect), self.Minv)[:, 0:3]

    def corners3d_to_img_boxes(self, corners3d):
        """(N,8,3) rect corners -> ((N,4) xyxy boxes, (N,8,2) corner pixels)  (:146-164)."""
        c = np.asarray(corners3d)
        hom = np.concatenate((c, np.ones((c.shape[0], 8, 1))), axis=2)
        p = np.matmul(hom, self.P2.T)
        x, y = p[:, :, 0] / p[:, :, 2], p[:, :, 1] / p[:, :, 2]
        boxes = np.stack((x.min(1), y.min(1), x.max(1), y.max(1)), axis=1)
        return boxes, np.stack((x, y), axis=2)

    def record(self):
        """48 float32: M43 | P2 | cu cv fu fv tx ty | Minv[:, :3] | pad."""
        r = np.zeros((48,), np.float32)
        r[0:12] = self.M43.reshape(-1)
        r[12:24] = self.P2.reshape(-1)
        r[24:30] = [self.cu, self.cv, self.fu, self.fv, self.tx, self.ty]
        r[30:42] = self.Minv[:, :3].reshape(-1)
        return r
