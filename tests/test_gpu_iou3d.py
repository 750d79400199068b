"""f-3: rotated BEV IoU / 3-D IoU / NMS on the GPU vs the CPU checker (same arithmetic text compiled by
gcc): IoU within 2e-6 (libm differences in cosf / sinf / atan2f), NMS keep lists identical."""
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


def test_bev_iou_and_overlap_match_checker():
    from dfu3d_amd.pcdet_kitti import iou3d_nms_utils as U
    from dfu3d_amd import stages as st
    rng = np.random.default_rng(21)
    for n, m in ((0, 4), (1, 1), (17, 33), (130, 257)):
        a, b = _boxes(rng, max(n, 1), 8.0)[:n], _boxes(rng, m, 8.0)
        if n and m:
            b[0] = a[0]                                     # identical pair
            if m > 1:
                b[1] = a[0]; b[1, 6] += np.float32(np.pi / 2)   # same box turned by 90 degrees
        ta, tb = torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV)
        iou = U.boxes_iou_bev(ta, tb).cpu().numpy()
        ov = st.boxes_bev(ta, tb, iou=False).cpu().numpy()
        exp_iou, exp_ov = I.boxes_bev(a, b), I.boxes_bev(a, b, iou=False)
        assert iou.shape == (n, m)
        np.testing.assert_allclose(iou, exp_iou, rtol=0, atol=2e-6)
        np.testing.assert_allclose(ov, exp_ov, rtol=2e-6, atol=2e-5)
        if n and m:
            assert abs(iou[0, 0] - 1.0) < 1e-5 and (exp_iou > 0).sum() >= min(m, 2)
            i3 = U.boxes_iou3d_gpu(ta, tb).cpu().numpy()
            np.testing.assert_allclose(i3, I.boxes_iou3d(a, b), rtol=0, atol=2e-6)


@pytest.mark.parametrize("n,thresh,pre", [(1, 0.1, None), (63, 0.1, None), (64, 0.01, None), (65, 0.3, None),
                                          (700, 0.1, None), (4096, 0.2, 3000), (2500, 0.7, None)])
def test_rotated_nms_matches_checker(n, thresh, pre):
    from dfu3d_amd.pcdet_kitti import iou3d_nms_utils as U
    rng = np.random.default_rng(100 + n)
    boxes = _boxes(rng, n, 4.0 * np.sqrt(n) ** 0.5 + 4.0)
    scores = rng.permutation(n).astype(np.float32) / n          # distinct -> the order is unambiguous
    sel, _ = U.nms_gpu(torch.from_numpy(boxes).to(DEV), torch.from_numpy(scores).to(DEV), thresh, pre_maxsize=pre)
    exp = I.nms(boxes, scores, thresh, pre_maxsize=pre)
    got = sel.cpu().numpy()
    if not np.array_equal(got, exp):
        # a pair whose IoU sits within libm noise of the threshold may legitimately flip; anything else is a bug
        iou = I.boxes_bev(boxes, boxes)
        near = np.abs(iou - thresh) < 3e-6
        assert near.any(), (n, len(got), len(exp))
        pytest.skip("keep lists differ only through a pair at the threshold (|iou - thresh| < 3e-6)")
    assert 0 < len(got) <= n
