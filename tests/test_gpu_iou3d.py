"""f-3 on the GPU against the independent float64 checker (oracle/iou3d_oracle.py: world-frame polygon
clipping, no code or formulation shared with the kernels): overlap / IoU within 1e-5 absolute, NMS keep lists identical
unless a pair sits within float32 noise of the threshold."""
import numpy as np
import pytest
import torch

from oracle import iou3d_oracle as I

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _boxes(rng, n, spread=20.0):
    b = np.zeros((n, 7), np.float32)
    b[:, :2] = rng.uniform(-spread, spread, (n, 2))
    b[:, 2] = rng.uniform(-2, 0, n)
    b[:, 3:6] = rng.uniform(0.6, 6.0, (n, 3))
    b[:, 6] = rng.uniform(-6.5, 6.5, n)
    return b


def test_bev_overlap_iou_and_volume_iou_match_the_checker():
    from dfu3d_amd.pcdet_kitti import iou3d_nms_utils as U
    from dfu3d_amd import stages as st
    rng = np.random.default_rng(21)
    for n, m in ((0, 4), (1, 1), (17, 33), (130, 257)):
        a, b = _boxes(rng, max(n, 1), 8.0)[:n], _boxes(rng, m, 8.0)
        if n and m:
            b[0] = a[0]                                     # identical pair
            if m > 1:
                b[1] = a[0]; b[1, 6] += np.float32(np.pi / 2)   # same box turned by 90 degrees
            if m > 2:
                b[2] = a[0]; b[2, 0] += a[0, 3]                 # pushed along x by its own length (heading aside)
        ta, tb = torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV)
        iou = U.boxes_iou_bev(ta, tb).cpu().numpy()
        ov = st.boxes_bev(ta, tb, iou=False).cpu().numpy()
        exp_iou, exp_ov = I.boxes_bev(a, b), I.boxes_bev(a, b, iou=False)
        assert iou.shape == (n, m)
        np.testing.assert_allclose(iou, exp_iou, rtol=0, atol=1e-5)
        np.testing.assert_allclose(ov, exp_ov, rtol=1e-5, atol=1e-4)
        if n and m:
            assert abs(iou[0, 0] - 1.0) < 1e-5 and (exp_iou > 0).sum() >= min(m, 2)
            i3 = U.boxes_iou3d_gpu(ta, tb).cpu().numpy()
            np.testing.assert_allclose(i3, I.boxes_iou3d(a, b), rtol=0, atol=1e-5)
            k = min(n, m)
            al = U.boxes_aligned_iou3d_gpu(ta[:k], tb[:k]).cpu().numpy()
            assert al.shape == (k, 1)
            np.testing.assert_allclose(al[:, 0], np.diag(I.boxes_iou3d(a[:k], b[:k])), rtol=0, atol=1e-5)


def test_far_from_the_origin_and_degenerate_boxes():
    """Box centres at hundreds of metres (float32 has ~3e-5 m resolution there: the kernels work relative to one box),
    zero-size boxes, and boxes that only touch."""
    from dfu3d_amd import stages as st
    rng = np.random.default_rng(5)
    a = _boxes(rng, 64, 3.0)
    b = _boxes(rng, 64, 3.0)
    a[:, :2] += np.float32(400.0); b[:, :2] += np.float32(400.0)
    b[0] = a[0]
    a[1, 3] = 0.0                                           # degenerate: no area
    b[2] = a[2]; b[2, 0] += a[2, 3] * np.cos(a[2, 6]); b[2, 1] += a[2, 3] * np.sin(a[2, 6])   # touching along an edge
    ov = st.boxes_bev(torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV), iou=False).cpu().numpy()
    exp = I.boxes_bev(a, b, iou=False)
    np.testing.assert_allclose(ov, exp, rtol=2e-5, atol=2e-3)       # 400 m * 2^-24 * perimeter
    assert (ov[1] < 1e-6).all() and ov[2, 2] < 2e-3            # a zero-width box has no area (up to float32 dust)
    assert np.isfinite(ov).all()


@pytest.mark.parametrize("criterion", [-1, 0, 1, 2])
def test_rotate_iou_eval_mirror(criterion):
    """The AP evaluator's entry point (numba-CUDA in the reference): NumPy in, NumPy float32 out."""
    from dfu3d_amd.pcdet_kitti.rotate_iou import rotate_iou_gpu_eval
    rng = np.random.default_rng(31 + criterion)
    n, k = 77, 130
    q = np.zeros((n, 5), np.float32); r = np.zeros((k, 5), np.float32)
    q[:, :2] = rng.uniform(-12, 12, (n, 2)); r[:, :2] = rng.uniform(-12, 12, (k, 2))
    q[:, 2:4] = rng.uniform(0.5, 5, (n, 2)); r[:, 2:4] = rng.uniform(0.5, 5, (k, 2))
    q[:, 4] = rng.uniform(-3.2, 3.2, n); r[:, 4] = rng.uniform(-3.2, 3.2, k)
    r[:5] = q[:5]
    got = rotate_iou_gpu_eval(q, r, criterion)
    assert got.dtype == np.float32 and got.shape == (n, k)
    exp = I.rotate_iou_eval(q, r, criterion)
    np.testing.assert_allclose(got, exp, rtol=1e-5, atol=1e-5 if criterion != 2 else 1e-4)
    assert (exp > 0).sum() > 50
    assert rotate_iou_gpu_eval(q[:0], r, criterion).shape == (0, k)


@pytest.mark.parametrize("n,thresh,pre,normal", [(1, 0.1, None, False), (63, 0.1, None, False), (64, 0.01, None, False),
                                                 (65, 0.3, None, False), (700, 0.1, None, False),
                                                 (4096, 0.2, 3000, False), (2500, 0.7, None, False),
                                                 (900, 0.25, None, True)])
def test_nms_matches_the_checker(n, thresh, pre, normal):
    from dfu3d_amd.pcdet_kitti import iou3d_nms_utils as U
    rng = np.random.default_rng(100 + n)
    boxes = _boxes(rng, n, 4.0 * np.sqrt(n) ** 0.5 + 4.0)
    scores = rng.permutation(n).astype(np.float32) / n          # distinct -> the order is unambiguous
    tb, ts = torch.from_numpy(boxes).to(DEV), torch.from_numpy(scores).to(DEV)
    if normal:
        sel, _ = U.nms_normal_gpu(tb, ts, thresh)
    else:
        sel, _ = U.nms_gpu(tb, ts, thresh, pre_maxsize=pre)
    exp, iou = I.nms(boxes, scores, thresh, pre_maxsize=pre, normal=normal)
    got = sel.cpu().numpy()
    assert 0 < len(got) <= n
    if not np.array_equal(got, exp):
        # a pair whose IoU sits within float32 noise of the threshold may legitimately flip; anything else is a bug
        near = np.abs(iou - thresh) < 1e-5
        assert near.any(), (n, len(got), len(exp))
        pytest.skip("keep lists differ only through a pair at the threshold (|iou - thresh| < 1e-5)")
