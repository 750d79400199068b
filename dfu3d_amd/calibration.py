"""Host mirror of the reference's calibration object
(tools/PENet/dataloaders/calibration_kitti.py:23-144): same constructor, same
attribute names.  The point transforms themselves run on the GPU
(csrc/common.hpp); this class only parses the KITTI calib text and packs the
48-float per-view record the kernels read (include/dfu3d.h).
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


class Calibration(object):
    """calibration_kitti.py:62-79 (constructor) + the packed GPU record."""

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

    def record(self):
        """48 float32: M43 | P2 | cu cv fu fv tx ty | Minv[:, :3] | pad."""
        r = np.zeros((48,), np.float32)
        r[0:12] = self.M43.reshape(-1)
        r[12:24] = self.P2.reshape(-1)
        r[24:30] = [self.cu, self.cv, self.fu, self.fv, self.tx, self.ty]
        r[30:42] = self.Minv[:, :3].reshape(-1)
        return r
