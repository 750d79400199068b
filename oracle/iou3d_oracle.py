"""CPU checker for SURVEY.md §8 row f-3 (rotated BEV IoU / NMS) -- TEST INFRASTRUCTURE ONLY.

Only tests/ and the cpu_baseline leg of tools/bench_iou.py may import this module.
The arithmetic (pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu:14-234, == iou3d_cpu.cpp:38-234) is the shared text
dfu3d_amd/csrc/iou_common.inc compiled by gcc into oracle/libdfu3d_oracle.so (oracle/csrc/iou3d_oracle.c); the HIP
library compiles the same text with hipcc, so GPU-vs-oracle tests isolate compiler / libm differences.
Parity: UNPINNED against the reference -- its CPU file includes <cuda.h> and cannot be compiled here, and it ships no
fixtures.  The restatement itself is therefore checked against GEOMETRY (tests/test_oracle_iou3d.py: analytic overlaps and
an independent float64 polygon clipper), and the host-side wrappers mirror pcdet/ops/iou3d_nms/iou3d_nms_utils.py.
"""
import ctypes

import numpy as np

from . import build as _build

_lib = None


def _L():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(_build.build())
        _lib.orc_boxes_bev.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                       ctypes.c_int]
        _lib.orc_nms.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
        _lib.orc_nms.restype = ctypes.c_int
    return _lib


def _boxes(b):
    b = np.ascontiguousarray(b, np.float32)
    assert b.ndim == 2 and b.shape[1] == 7
    return b


def boxes_bev(boxes_a, boxes_b, iou=True):
    """(N,7),(M,7) -> (N,M) float32: BEV IoU (iou3d_nms_kernel.cu:226-234) or overlap area (:117-224)."""
    a, b = _boxes(boxes_a), _boxes(boxes_b)
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    if a.shape[0] and b.shape[0]:
        _L().orc_boxes_bev(a.ctypes.data, a.shape[0], b.ctypes.data, b.shape[0], out.ctypes.data, 1 if iou else 0)
    return out


def boxes_iou3d(boxes_a, boxes_b):
    """iou3d_nms_utils.py:48-81 (float32 torch arithmetic restated in NumPy float32)."""
    a, b = _boxes(boxes_a), _boxes(boxes_b)
    f = np.float32
    a_max, a_min = (a[:, 2] + a[:, 5] / f(2)).reshape(-1, 1), (a[:, 2] - a[:, 5] / f(2)).reshape(-1, 1)
    b_max, b_min = (b[:, 2] + b[:, 5] / f(2)).reshape(1, -1), (b[:, 2] - b[:, 5] / f(2)).reshape(1, -1)
    ov = boxes_bev(a, b, iou=False)
    oh = np.maximum(np.minimum(a_max, b_max) - np.maximum(a_min, b_min), f(0))
    o3 = ov * oh
    va = (a[:, 3] * a[:, 4] * a[:, 5]).reshape(-1, 1)
    vb = (b[:, 3] * b[:, 4] * b[:, 5]).reshape(1, -1)
    return o3 / np.maximum(va + vb - o3, f(1e-6))


def nms(boxes, scores, thresh, pre_maxsize=None):
    """iou3d_nms_utils.py:120-135: indices (into the input) of the kept boxes, best score first."""
    order = np.argsort(-np.asarray(scores, np.float32), kind="stable")
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    b = _boxes(np.asarray(boxes)[order])
    keep = np.zeros(b.shape[0], np.int64)
    n = _L().orc_nms(b.ctypes.data, b.shape[0], ctypes.c_float(thresh), keep.ctypes.data) if b.shape[0] else 0
    return order[keep[:n]]
