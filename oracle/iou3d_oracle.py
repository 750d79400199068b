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


# ---- the reference's OWN overlap construction, for measuring how far it is from the exact area -------------------------
def overlap_reference_margin(box_a, box_b, margin=1e-2, eps=1e-8):
    """Restatement (test infrastructure; float32 like the kernel) of the reference's `box_overlap`
    (pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu:108-222): the polygon is assembled from (i) the intersections of the
    4 x 4 edge pairs that cross strictly (`intersection`, :61-92), (ii) the corners of either box that lie inside
    the other one ENLARGED BY `margin` = 1e-2 on every side (`check_in_box2d`, :49-59), then sorted by angle about their
    mean (bubble sort with a strict `>`, :189-198) and summed as a triangle fan (:208-211).  A corner up to 1 cm OUTSIDE
    the other box therefore joins the polygon: the result can exceed the true intersection by a sliver (bounded by
    margin x the perimeter in play), and for boxes that share edges it can also fall short (collinear edges do not
    "cross").  One pair per call: boxes (7,) [x y z dx dy dz heading].  Used only by tests that count how many
    suppression decisions differ between this construction and the exact area the product computes."""
    f = np.float32
    a, b = np.asarray(box_a, f), np.asarray(box_b, f)

    def corners(bx):
        hx, hy = bx[3] / f(2), bx[4] / f(2)
        pts = np.array([[bx[0] - hx, bx[1] - hy], [bx[0] + hx, bx[1] - hy], [bx[0] + hx, bx[1] + hy], [bx[0] - hx, bx[1] + hy]], f)
        c, s = f(np.cos(bx[6])), f(np.sin(bx[6]))
        out = np.empty((5, 2), f)
        for k in range(4):
            dx, dy = pts[k, 0] - bx[0], pts[k, 1] - bx[1]
            out[k, 0] = dx * c + dy * (-s) + bx[0]
            out[k, 1] = dx * s + dy * c + bx[1]
        out[4] = out[0]
        return out

    def cross3(p1, p2, p0):
        return (p1[0] - p0[0]) * (p2[1] - p0[1]) - (p2[0] - p0[0]) * (p1[1] - p0[1])

    def intersection(p1, p0, q1, q0):
        if not (min(p0[0], p1[0]) <= max(q0[0], q1[0]) and min(q0[0], q1[0]) <= max(p0[0], p1[0]) and
                min(p0[1], p1[1]) <= max(q0[1], q1[1]) and min(q0[1], q1[1]) <= max(p0[1], p1[1])):
            return None
        s1, s2, s3, s4 = cross3(q0, p1, p0), cross3(p1, q1, p0), cross3(p0, q1, q0), cross3(q1, p1, q0)
        if not (s1 * s2 > 0 and s3 * s4 > 0):
            return None
        s5 = cross3(q1, p1, p0)
        if abs(s5 - s1) > eps:
            return np.array([(s5 * q0[0] - s1 * q1[0]) / (s5 - s1), (s5 * q0[1] - s1 * q1[1]) / (s5 - s1)], f)
        a0, b0, c0 = p0[1] - p1[1], p1[0] - p0[0], p0[0] * p1[1] - p1[0] * p0[1]
        a1, b1, c1 = q0[1] - q1[1], q1[0] - q0[0], q0[0] * q1[1] - q1[0] * q0[1]
        D = a0 * b1 - a1 * b0
        return np.array([(b0 * c1 - b1 * c0) / D, (a1 * c0 - a0 * c1) / D], f)

    def inside(bx, p):
        c, s = f(np.cos(-bx[6])), f(np.sin(-bx[6]))
        rx = (p[0] - bx[0]) * c + (p[1] - bx[1]) * (-s)
        ry = (p[0] - bx[0]) * s + (p[1] - bx[1]) * c
        return abs(rx) < bx[3] / f(2) + f(margin) and abs(ry) < bx[4] / f(2) + f(margin)

    with np.errstate(all="ignore"):
        ca, cb = corners(a), corners(b)
        pts = []
        for i in range(4):
            for j in range(4):
                p = intersection(ca[i + 1], ca[i], cb[j + 1], cb[j])
                if p is not None:
                    pts.append(p)
        for k in range(4):
            if inside(a, cb[k]):
                pts.append(cb[k].copy())
            if inside(b, ca[k]):
                pts.append(ca[k].copy())
        cnt = len(pts)
        if cnt == 0:
            return 0.0
        P = np.array(pts, f)
        ctr = P.sum(0) / f(cnt)
        ang = [f(np.arctan2(p[1] - ctr[1], p[0] - ctr[0])) for p in P]
        for j in range(cnt - 1):                       # the kernel's bubble sort (strict >: equal angles keep their order)
            for i in range(cnt - j - 1):
                if ang[i] > ang[i + 1]:
                    P[[i, i + 1]] = P[[i + 1, i]]
                    ang[i], ang[i + 1] = ang[i + 1], ang[i]
        area = f(0)
        for k in range(cnt - 1):
            u, v = P[k] - P[0], P[k + 1] - P[0]
            area += u[0] * v[1] - u[1] * v[0]
        return float(abs(area) / f(2))


def iou_bev_reference_margin(box_a, box_b):
    """iou3d_nms_kernel.cu:224-231 on top of overlap_reference_margin."""
    sa, sb = float(np.float32(box_a[3]) * np.float32(box_a[4])), float(np.float32(box_b[3]) * np.float32(box_b[4]))
    s = overlap_reference_margin(box_a, box_b)
    return s / max(sa + sb - s, 1e-8)
