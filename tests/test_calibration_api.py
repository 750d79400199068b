"""The PRODUCT's Calibration class (dfu3d_amd.calibration / dfu3d_amd.penet.calibration_kitti,
the module INTEGRATION.md §1 swaps in for tools/PENet/dataloaders/calibration_kitti.py) against
golden G1, i.e. against the reference's own class: constructor attributes and the six methods the
reference's callers use (vis_utils.py:152-154, my_loader.py:518,535-536,679).  CPU only; the oracle
is not involved."""
import numpy as np
import pytest

from dfu3d_amd.calibration import fma_f32, _round_f32
from dfu3d_amd.penet import calibration_kitti


def _calib(tmp_path, text_u8):
    p = tmp_path / "000000.txt"
    p.write_bytes(bytes(text_u8))
    return calibration_kitti.Calibration(str(p))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_product_calibration_matches_reference_g1(golden_dir, tmp_path, tag):
    g = np.load(golden_dir + "/g1_calib.npz")
    c = _calib(tmp_path, g[tag + "_text"])
    assert np.array_equal(c.P2, g[tag + "_P2"]) and np.array_equal(c.R0, g[tag + "_R0"])
    assert np.array_equal(c.V2C, g[tag + "_V2C"])
    assert np.array_equal(np.array([c.cu, c.cv, c.fu, c.fv, c.tx, c.ty], np.float32), g[tag + "_scal"])
    # float32 in -> float32 out, bit-exact (the reference's BLAS chain)
    rect = c.lidar_to_rect(g[tag + "_pts32"])
    assert rect.dtype == np.float32 and np.array_equal(rect, g[tag + "_rect32"])
    img, dep = c.rect_to_img(rect)
    assert img.dtype == np.float32 and dep.dtype == np.float32
    assert np.array_equal(img, g[tag + "_img32"]) and np.array_equal(dep, g[tag + "_dep32"])
    img_b, dep_b = c.lidar_to_img(g[tag + "_pts32"])
    assert np.array_equal(img_b, g[tag + "_img_b"]) and np.array_equal(dep_b, g[tag + "_dep_b"])
    # float64 points stay float64
    r64 = c.lidar_to_rect(g[tag + "_pts64"])
    assert r64.dtype == np.float64
    np.testing.assert_allclose(r64, g[tag + "_rect64"], rtol=1e-13, atol=1e-12)
    # int64 pixels + float32 depth -> float64 (my_loader.py:535-536)
    prect = c.img_to_rect(g[tag + "_u"], g[tag + "_v"], g[tag + "_d"])
    assert prect.dtype == np.float64 and np.array_equal(prect, g[tag + "_prect"])
    plid = c.rect_to_lidar(prect)
    assert plid.shape == (256, 3)
    np.testing.assert_allclose(plid, g[tag + "_plid"], rtol=1e-13, atol=1e-12)
    hom = c.cart_to_hom(g[tag + "_pts32"])
    assert hom.dtype == np.float32 and hom.shape == (256, 4) and (hom[:, 3] == 1).all()
    assert c.cart_to_hom(g[tag + "_pts64"]).dtype == np.float64


def test_fov_prefilter_of_vis_utils_through_the_swapped_module(golden_dir, tmp_path):
    """vis_utils.py:152-154 as the reference writes it, with the swapped-in class:
    lidar[get_fov_flag(calib.lidar_to_rect(lidar[:, 0:3]), image.shape, calib)] == the reference's flag (G7)."""
    for tag in ("dense", "dense2"):
        g = np.load(golden_dir + "/g7_%s.npz" % tag)
        c = _calib(tmp_path, g["calib_text"])
        pts_rect = c.lidar_to_rect(g["lidar_all"][:, 0:3])
        pts_img, depth = c.rect_to_img(pts_rect)                        # get_fov_flag, vis_utils.py:108-123
        flag = (pts_img[:, 0] >= 0) & (pts_img[:, 0] < 1600) & (pts_img[:, 1] >= 0) & (pts_img[:, 1] < 900) & (depth >= 0)
        assert np.array_equal(flag, g["fov"])


def test_generate_anns_centre_transform(golden_dir, tmp_path):
    """my_loader.py:677-679: one fp64 box centre through lidar_to_rect (float64 path, single row)."""
    g = np.load(golden_dir + "/g1_calib.npz")
    c = _calib(tmp_path, g["b_text"])
    p = g["b_pts64"][3:4]
    np.testing.assert_allclose(c.lidar_to_rect(p), g["b_rect64"][3:4], rtol=1e-13, atol=1e-12)


def test_fma_f32_is_correctly_rounded():
    from fractions import Fraction
    rng = np.random.default_rng(11)
    a = rng.normal(0, 10, 600).astype(np.float32)
    b = rng.normal(0, 10, 600).astype(np.float32)
    # near-cancellation makes the double rounding of a naive float64 evaluation visible
    c = (-(a.astype(np.float64) * b.astype(np.float64))).astype(np.float32) + rng.normal(0, 1e-4, 600).astype(np.float32)
    got = fma_f32(a, b, c)
    for i in range(600):
        assert got[i] == _round_f32(Fraction(float(a[i])) * Fraction(float(b[i])) + Fraction(float(c[i])))


def test_corners3d_to_img_boxes(golden_dir, tmp_path):
    g = np.load(golden_dir + "/g1_calib.npz")
    c = _calib(tmp_path, g["a_text"])
    rng = np.random.default_rng(2)
    corners = rng.normal(0, 2, (5, 8, 3)) + np.array([0.0, 0.0, 20.0])
    boxes, pix = c.corners3d_to_img_boxes(corners)
    assert boxes.shape == (5, 4) and pix.shape == (5, 8, 2)
    uv, _ = c.rect_to_img(corners.reshape(-1, 3))
    np.testing.assert_allclose(pix.reshape(-1, 2), uv, rtol=1e-12)
    assert (boxes[:, 0] <= boxes[:, 2]).all() and (boxes[:, 1] <= boxes[:, 3]).all()
