"""f-4 oracle (la_sampling of the GT-sampling augmentor) against hand-made cases.  CPU only."""
import numpy as np

from oracle import la_sampling_oracle as LA


def _sph(theta, fan, r=10.0):
    """float32 point whose spherical angles are (about) theta, fan."""
    return np.array([r * np.sin(theta) * np.cos(fan), r * np.sin(theta) * np.sin(fan), r * np.cos(theta), 0.5], np.float32)


def test_bins_keep_their_min_theta_point_in_first_seen_order():
    vr, hr = 0.006, 0.003
    # six bins; the first bin gets three points, the later of which has the smallest theta
    pts = np.stack([_sph(1.5012, 0.0101), _sph(1.5090, 0.0301), _sph(1.5001, 0.0102), _sph(1.5200, 0.0701),
                    _sph(1.5030, 0.0115), _sph(1.5400, -0.0401), _sph(1.5610, 0.1001), _sph(1.5800, 0.2001)])
    pts[:, 3] = np.arange(len(pts), dtype=np.float32)
    out, idx = LA.la_sampling(pts, vr, hr, return_index=True)
    assert idx.tolist() == [2, 1, 3, 5, 6, 7]           # bin of point 0 first (its min-theta member is point 2), then first-seen
    assert np.array_equal(out, pts[idx])
    # fewer than five bins: the object comes back unchanged
    few = pts[[0, 2, 4, 1]]
    assert LA.la_sampling(few, vr, hr) is few or np.array_equal(LA.la_sampling(few, vr, hr), few)


def test_key_is_the_string_of_the_float32_quotient():
    # negative fan: floor division rounds toward -inf; +-0.0 are different strings, every NaN is 'nan'
    pts = np.stack([_sph(1.2, -0.0001), _sph(1.2, 0.0001), _sph(1.3, -0.2), _sph(1.4, 0.2), _sph(1.1, 0.5), _sph(1.0, 0.7)])
    sp = LA.to_sphere_coords(pts)
    q = sp[:, 2] // 0.003
    assert q.dtype == np.float32 and q[0] == -1.0 and q[1] == 0.0
    out = LA.la_sampling(pts)
    assert len(out) == 6
    z = np.zeros((6, 4), np.float32)                     # r = 0: theta and fan are NaN -> one bin 'nan_nan'
    assert np.array_equal(LA.la_sampling(z), z)          # one bin < 5 -> unchanged
