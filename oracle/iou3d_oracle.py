"""CPU checker for SURVEY.md §8 row f-3 (overlap of rotated boxes, IoU criteria, NMS) -- TEST INFRASTRUCTURE ONLY.

Only tests/ and the cpu_baseline leg of tools/bench_iou.py may import this module.

It shares NO text and no formulation with the product's kernels (dfu3d_amd/csrc/iou_stage.hip clips in the frame of one
box, float32, polygon in LDS): here both rectangles stay in WORLD coordinates, float64, and rectangle A is clipped by the
four directed edge lines of rectangle B (convex-polygon clipping, vectorised over pairs with NumPy).  What is computed is
the exact area of the intersection polygon.
Parity: UNPINNED against the reference (pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu:14-234, iou3d_cpu.cpp include
<cuda.h> / torch headers and cannot be compiled here; the reference ships no fixtures).  Its kernels are known to
over-estimate the overlap by a sliver when a corner lies within their 1e-2 margin outside the other box; this oracle and
the product do not.  The oracle itself is checked against analytic overlaps (tests/test_oracle_iou3d.py).
"""
import numpy as np


def _corners7(b):
    """(n,7) [x y z dx dy dz heading] -> (n,4,2) counter-clockwise corners, float64 (heading turns counter-clockwise)."""
    b = np.asarray(b, np.float64)
    c, s = np.cos(b[:, 6]), np.sin(b[:, 6])
    hx, hy = b[:, 3] / 2, b[:, 4] / 2
    sx = np.array([1.0, -1.0, -1.0, 1.0])
    sy = np.array([1.0, 1.0, -1.0, -1.0])
    x = b[:, 0:1] + sx * hx[:, None] * c[:, None] - sy * hy[:, None] * s[:, None]
    y = b[:, 1:2] + sx * hx[:, None] * s[:, None] + sy * hy[:, None] * c[:, None]
    return np.stack([x, y], -1)


def _corners5(b):
    """(n,5) [cx cy w h angle] of rotate_iou.py (its corner formula turns the box clockwise by `angle`)."""
    b = np.asarray(b, np.float64)
    b7 = np.zeros((b.shape[0], 7))
    b7[:, 0:2], b7[:, 3:5], b7[:, 6] = b[:, 0:2], b[:, 2:4], -b[:, 4]
    return _corners7(b7)


def _clip_pairs(PA, PB):
    """Areas of PA[i] n PB[i] for (p,4,2) counter-clockwise quadrilaterals (float64)."""
    p = PA.shape[0]
    V = np.zeros((p, 8, 2))
    V[:, :4] = PA
    n = np.full(p, 4)
    idx = np.arange(8)
    for e in range(4):
        a, b = PB[:, e], PB[:, (e + 1) % 4]
        d = b - a                                                    # inside = left of a -> b
        nxt = np.where(idx[None, :] + 1 < n[:, None], idx[None, :] + 1, 0)
        Vn = np.take_along_axis(V, nxt[:, :, None], 1)
        g0 = d[:, None, 0] * (V[:, :, 1] - a[:, None, 1]) - d[:, None, 1] * (V[:, :, 0] - a[:, None, 0])
        g1 = d[:, None, 0] * (Vn[:, :, 1] - a[:, None, 1]) - d[:, None, 1] * (Vn[:, :, 0] - a[:, None, 0])
        live = idx[None, :] < n[:, None]
        keep = live & (g0 >= 0)
        cross = live & ((g0 >= 0) != (g1 >= 0))
        with np.errstate(all="ignore"):
            t = np.where(cross, g0 / (g0 - g1), 0.0)
        X = V + t[:, :, None] * (Vn - V)
        # output order per input vertex k: [V_k if kept] then [crossing if any]
        cnt = keep.astype(np.int64) + cross.astype(np.int64)
        off = np.cumsum(cnt, 1) - cnt
        out = np.zeros((p, 9, 2))
        rows = np.repeat(np.arange(p)[:, None], 8, 1)
        out[rows[keep], off[keep]] = V[keep]
        pos = off + keep.astype(np.int64)
        out[rows[cross], pos[cross]] = X[cross]
        n = cnt.sum(1)
        V = out[:, :8]
    x, y = V[:, :, 0], V[:, :, 1]
    nxt = np.where(idx[None, :] + 1 < n[:, None], idx[None, :] + 1, 0)
    xn, yn = np.take_along_axis(x, nxt, 1), np.take_along_axis(y, nxt, 1)
    live = idx[None, :] < n[:, None]
    area = 0.5 * np.abs(np.where(live, x * yn - xn * y, 0.0).sum(1))
    return np.where(n >= 3, area, 0.0)


def _overlap_matrix(CA, CB):
    """(n,4,2),(m,4,2) -> (n,m) float64 areas; pairs whose circumscribed circles are apart are skipped (exactly 0)."""
    n, m = CA.shape[0], CB.shape[0]
    out = np.zeros((n, m))
    if n == 0 or m == 0:
        return out
    ca, cb = CA.mean(1), CB.mean(1)
    ra = np.linalg.norm(CA[:, 0] - ca, axis=1)
    rb = np.linalg.norm(CB[:, 0] - cb, axis=1)
    for i0 in range(0, n, 256):                       # bounded memory
        d = np.linalg.norm(ca[i0:i0 + 256, None, :] - cb[None, :, :], axis=2)
        ii, jj = np.nonzero(d <= (ra[i0:i0 + 256, None] + rb[None, :]) * (1 + 1e-9) + 1e-12)
        if ii.size:
            out[ii + i0, jj] = _clip_pairs(CA[ii + i0], CB[jj])
    return out


def boxes_bev(boxes_a, boxes_b, iou=True):
    """(N,7),(M,7) -> (N,M) float64: BEV IoU or overlap area of the exact intersection polygons."""
    a, b = np.asarray(boxes_a, np.float64).reshape(-1, 7), np.asarray(boxes_b, np.float64).reshape(-1, 7)
    ov = _overlap_matrix(_corners7(a), _corners7(b))
    if not iou:
        return ov
    sa, sb = (a[:, 3] * a[:, 4])[:, None], (b[:, 3] * b[:, 4])[None, :]
    return ov / np.maximum(sa + sb - ov, 1e-8)


def boxes_iou3d(boxes_a, boxes_b):
    """3-D IoU: BEV overlap x overlap of the z extents over the union of the volumes (iou3d_nms_utils.py:48-81)."""
    a, b = np.asarray(boxes_a, np.float64).reshape(-1, 7), np.asarray(boxes_b, np.float64).reshape(-1, 7)
    ov = boxes_bev(a, b, iou=False)
    top = np.minimum((a[:, 2] + a[:, 5] / 2)[:, None], (b[:, 2] + b[:, 5] / 2)[None, :])
    bot = np.maximum((a[:, 2] - a[:, 5] / 2)[:, None], (b[:, 2] - b[:, 5] / 2)[None, :])
    inter = ov * np.maximum(top - bot, 0.0)
    vol = (a[:, 3] * a[:, 4] * a[:, 5])[:, None] + (b[:, 3] * b[:, 4] * b[:, 5])[None, :]
    return inter / np.maximum(vol - inter, 1e-6)


def rotate_iou_eval(boxes, query_boxes, criterion=-1):
    """(N,5),(K,5) [cx cy w h angle] -> (N,K): rotate_iou.py:247-255 criteria on the exact overlap."""
    a, q = np.asarray(boxes, np.float64).reshape(-1, 5), np.asarray(query_boxes, np.float64).reshape(-1, 5)
    ov = _overlap_matrix(_corners5(a), _corners5(q))
    sa, sq = (a[:, 2] * a[:, 3])[:, None], (q[:, 2] * q[:, 3])[None, :]
    if criterion == -1:
        return ov / np.maximum(sa + sq - ov, 1e-8)
    if criterion == 0:
        return ov / np.maximum(sa, 1e-8)
    if criterion == 1:
        return ov / np.maximum(sq, 1e-8)
    return ov


def nms(boxes, scores, thresh, pre_maxsize=None, iou=None, normal=False):
    """Greedy suppression in score order -> (indices into the input of the kept boxes, best first; the IoU matrix
    of the score-sorted boxes).  normal=True: axis-aligned IoU, headings ignored."""
    order = np.argsort(-np.asarray(scores, np.float64), kind="stable")
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    b = np.asarray(boxes, np.float64)[order]
    if iou is None:
        if normal:
            x0, x1 = b[:, 0] - b[:, 3] / 2, b[:, 0] + b[:, 3] / 2
            y0, y1 = b[:, 1] - b[:, 4] / 2, b[:, 1] + b[:, 4] / 2
            w = np.maximum(np.minimum(x1[:, None], x1[None, :]) - np.maximum(x0[:, None], x0[None, :]), 0)
            h = np.maximum(np.minimum(y1[:, None], y1[None, :]) - np.maximum(y0[:, None], y0[None, :]), 0)
            ov = w * h
            s = b[:, 3] * b[:, 4]
            iou = ov / np.maximum(s[:, None] + s[None, :] - ov, 1e-8)
        else:
            iou = boxes_bev(b, b)
    n = b.shape[0]
    dead = np.zeros(n, bool)
    keep = []
    for i in range(n):
        if dead[i]:
            continue
        keep.append(i)
        dead[i + 1:] |= iou[i, i + 1:] > thresh
    return order[np.array(keep, np.int64)], iou
