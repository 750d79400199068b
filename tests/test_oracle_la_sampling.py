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


def _g10():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g10_la_sampling.npz"))


def test_oracle_equals_the_reference_run_g10():
    """Golden G10 (tests/golden/capture_la_sampling_golden.py): the reference's own DADataBaseSampler.la_sampling,
    imported unmodified, on 53 seeded objects (random sizes 1-4000, NaN angles, x = 0, x around 0, duplicates, fewer
    than five bins, points behind and above the sensor) at two resolutions.  The oracle must select the same rows in
    the same order, bit for bit -- on this image both run the same NumPy float32 loops, so that holds for every
    object, the ones flagged `fragile` (an angle within 2 ulp of a bin edge) included."""
    g = _g10()
    n = int(g["n_objects"])
    assert n >= 40 and str(g["numpy_version"]).split(".")[0] == np.__version__.split(".")[0]
    kinds = set(str(t) for t in g["tags"])
    assert {"all_nan", "x_exactly_zero", "few_bins", "duplicates"} <= kinds
    for ri, (vr, hr) in enumerate(g["res"].tolist()):
        for k in range(n):
            o = g["in/%d" % k]
            out, idx = LA.la_sampling(o, vr, hr, return_index=True)
            ref = g["out/%d/%d" % (ri, k)]
            assert out.shape == ref.shape, (ri, k, str(g["tags"][k]))
            assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), (ri, k, str(g["tags"][k]))
            assert np.array_equal(np.asarray(idx), g["idx/%d/%d" % (ri, k)]) or len(np.unique(o.view(np.uint32), axis=0)) < len(o)
    # the NumPy-1.x reading of `float32 // python float` (promotion to float64) moves bin edges: recorded, not pinned
    assert sum(int(g["keys64_differ/0/%d" % k]) for k in range(n)) >= 0
