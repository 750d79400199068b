"""f-3 oracle (rotated BEV IoU / NMS) against geometry: analytic overlaps and an independent float64
Sutherland-Hodgman clipper.  The reference's corner test has a 1e-2 margin (iou3d_nms_kernel.cu:53), so generic
(non-touching) configurations are used and areas agree to ~1e-4 relative."""
import numpy as np

from oracle import iou3d_oracle as I


def _corners(b):
    x, y, dx, dy, r = b[0], b[1], b[3], b[4], b[6]
    c, s = np.cos(r), np.sin(r)
    pts = np.array([[-dx / 2, -dy / 2], [dx / 2, -dy / 2], [dx / 2, dy / 2], [-dx / 2, dy / 2]])
    return pts @ np.array([[c, s], [-s, c]]) + [x, y]


def _clip_area(pa, pb):
    """Area of the intersection of two convex CCW polygons (Sutherland-Hodgman, float64)."""
    out = [tuple(p) for p in pa]
    for i in range(len(pb)):
        a, b = pb[i], pb[(i + 1) % len(pb)]
        inp, out = out, []
        if not inp:
            break
        side = lambda p: (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])
        for k in range(len(inp)):
            p, q = inp[k], inp[(k + 1) % len(inp)]
            sp, sq = side(p), side(q)
            if sp >= 0:
                out.append(p)
            if sp * sq < 0:
                t = sp / (sp - sq)
                out.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
    if len(out) < 3:
        return 0.0
    x, y = np.array(out).T
    return 0.5 * abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))


def test_analytic_cases():
    A = np.array([[0, 0, 0, 4, 2, 1.5, 0.0]], np.float32)
    assert abs(I.boxes_bev(A, A)[0, 0] - 1.0) < 1e-6                                   # identical
    B = np.array([[2, 0, 0, 4, 2, 1.5, 0.0]], np.float32)                             # shifted by half its length
    assert abs(I.boxes_bev(A, B, iou=False)[0, 0] - 4.0) < 1e-5
    assert abs(I.boxes_bev(A, B)[0, 0] - 4.0 / 12.0) < 1e-6
    C = np.array([[10, 10, 0, 4, 2, 1.5, 0.3]], np.float32)                            # disjoint
    assert I.boxes_bev(A, C)[0, 0] == 0.0
    S = np.array([[0, 0, 0, 2, 2, 1, 0.0]], np.float32)                                # square vs itself turned by 45 deg:
    T = np.array([[0, 0, 0, 2, 2, 1, np.pi / 4]], np.float32)                          # regular octagon, area 8(sqrt2-1)
    assert abs(I.boxes_bev(S, T, iou=False)[0, 0] - 8 * (np.sqrt(2) - 1)) < 1e-4
    small = np.array([[0.3, -0.2, 0, 1, 0.5, 1, 1.1]], np.float32)                     # contained
    assert abs(I.boxes_bev(A, small, iou=False)[0, 0] - 0.5) < 1e-5
    # 3-D IoU: same footprint, half the height overlapping
    U = np.array([[0, 0, 0.75, 4, 2, 1.5, 0.0]], np.float32)
    assert abs(I.boxes_iou3d(A, U)[0, 0] - (8 * 0.75) / (12 + 12 - 6)) < 1e-6


def test_random_pairs_against_polygon_clipping():
    rng = np.random.default_rng(3)
    n = 300
    a = np.zeros((n, 7), np.float32)
    b = np.zeros((n, 7), np.float32)
    a[:, :2] = rng.uniform(-3, 3, (n, 2)); b[:, :2] = a[:, :2] + rng.uniform(-2.5, 2.5, (n, 2))
    a[:, 3:5] = rng.uniform(0.8, 5, (n, 2)); b[:, 3:5] = rng.uniform(0.8, 5, (n, 2))
    a[:, 5] = b[:, 5] = 1.5
    a[:, 6] = rng.uniform(-3.2, 3.2, n); b[:, 6] = rng.uniform(-3.2, 3.2, n)
    ov = I.boxes_bev(a, b, iou=False)
    worst = 0.0
    for i in range(n):
        exp = _clip_area(_corners(a[i].astype(np.float64)), _corners(b[i].astype(np.float64)))
        # the 1e-2 corner margin can add at most a thin sliver along the perimeter
        tol = 2e-2 * (a[i, 3] + a[i, 4] + b[i, 3] + b[i, 4]) * 0.5 + 1e-4
        assert abs(ov[i, i] - exp) <= tol, (i, ov[i, i], exp)
        worst = max(worst, abs(ov[i, i] - exp))
    assert worst < 0.12


def test_nms_is_greedy_on_sorted_scores():
    rng = np.random.default_rng(4)
    n = 200
    boxes = np.zeros((n, 7), np.float32)
    boxes[:, :2] = rng.uniform(-10, 10, (n, 2))
    boxes[:, 3:6] = rng.uniform(1.5, 4.5, (n, 3))
    boxes[:, 6] = rng.uniform(-3, 3, n)
    scores = rng.permutation(n).astype(np.float32)
    keep = I.nms(boxes, scores, 0.1)
    iou = I.boxes_bev(boxes, boxes)
    order = np.argsort(-scores)
    kept, rem = [], set()
    for i in order:
        if i in rem:
            continue
        kept.append(i)
        for j in order:
            if scores[j] < scores[i] and iou[i, j] > 0.1:
                rem.add(j)
    assert keep.tolist() == kept and 10 < len(kept) < n
    assert I.nms(boxes, scores, 0.1, pre_maxsize=50).tolist() == [k for k in kept if k in set(order[:50])][:len(I.nms(boxes, scores, 0.1, pre_maxsize=50))]
