"""CPU tests of the oracle's Open3D voxel_down_sample restatement (my_loader0.py:734): known answers
computed by hand, the sequential accumulation order, and the DEFINED (first-seen) output order.  The Open3D leaf itself
is absent from the reference tree and from this image: parity unpinned, see the function's header."""
import numpy as np
import pytest

from oracle import penet_oracle as O


def test_known_answer_two_voxels():
    # voxel_min_bound = (0, 0, 0) - 0.025; points 0, 1, 3 fall into voxel (0, 0, 0) / (0, 0, 0) / (0, 0, 0), 2 and 4 into (20, 0, 0)
    pts = np.array([[0.0, 0.0, 0.0], [0.01, 0.01, 0.0], [1.0, 0.0, 0.0], [0.02, 0.0, 0.01], [1.01, 0.01, 0.0]])
    out = O.voxel_down_sample(pts, 0.05)
    exp = np.array([((pts[0] + pts[1]) + pts[3]) / 3.0, (pts[2] + pts[4]) / 2.0])
    assert np.array_equal(out, exp)


def test_first_seen_order_and_min_bound_shift():
    # the grid hangs on the minimum of the cloud, shifted by half a voxel: a point exactly at the minimum sits in the
    # MIDDLE of voxel 0, so +-0.02 around it stays in voxel 0 and +0.03 is voxel 1
    pts = np.array([[5.03, 0.0, 0.0], [5.0, 0.0, 0.0], [5.02, 0.0, 0.0], [9.0, 1.0, 1.0], [5.035, 0.0, 0.0]])
    out = O.voxel_down_sample(pts, 0.05)
    # voxels in the order of their first point: {0, 4} (index 1 along x), {1, 2} (index 0), {3}
    exp = np.array([(pts[0] + pts[4]) / 2.0, (pts[1] + pts[2]) / 2.0, pts[3]])
    assert np.array_equal(out, exp)


def test_sums_are_formed_in_input_order():
    # three values whose float64 sum depends on the order: (a + b) + c != a + (b + c)
    a, b, c = 0.1, 0.2, 0.3
    assert (a + b) + c != a + (b + c)
    pts = np.array([[a, 0, 0], [b, 0, 0], [c, 0, 0]]) * 1e-3 + np.array([7.0, 0, 0])
    out = O.voxel_down_sample(pts, 0.05)
    assert out.shape == (1, 3)
    assert out[0, 0] == ((pts[0, 0] + pts[1, 0]) + pts[2, 0]) / 3.0


def test_against_a_plain_python_dictionary():
    rng = np.random.default_rng(5)
    pts = np.cumsum(rng.normal(0, 0.02, (3000, 3)), 0)
    pts[rng.random(3000) < 0.02] += rng.uniform(-5, 5, 3)
    vs = 0.05
    minb = pts.min(0) - vs * 0.5
    acc, order = {}, []
    for p_ in pts:                                  # Open3D's loop, with a dictionary that remembers insertion order
        k = tuple(int(np.floor(v)) for v in (p_ - minb) / vs)
        if k not in acc:
            acc[k] = [np.zeros(3), 0]
            order.append(k)
        acc[k][0] = acc[k][0] + p_
        acc[k][1] += 1
    exp = np.array([acc[k][0] / float(acc[k][1]) for k in order])
    out = O.voxel_down_sample(pts, vs)
    assert np.array_equal(out, exp) and 100 < len(exp) < 3000


def test_edge_cases():
    assert O.voxel_down_sample(np.zeros((0, 3)), 0.05).shape == (0, 3)
    one = np.array([[1.0, 2.0, 3.0]])
    assert np.array_equal(O.voxel_down_sample(one, 0.05), one)
    dup = np.repeat(one, 7, 0)
    assert np.allclose(O.voxel_down_sample(dup, 0.05), one, rtol=0, atol=1e-15) and O.voxel_down_sample(dup, 0.05).shape == (1, 3)
    with pytest.raises(ValueError):
        O.voxel_down_sample(one, 0.0)


def test_statistical_pair_runs_on_the_centroids():
    """instance_points with stat_filter: voxel_down_sample -> remove_statistical_outlier -> fuse (my_loader0.py:734-735)."""
    rng = np.random.default_rng(9)
    lid = rng.normal(0, 0.5, (50, 3))
    pse = np.concatenate([lid[rng.integers(0, 50, 600)] + rng.normal(0, 0.03, (600, 3)), rng.uniform(-4, 4, (30, 3))])
    p = O.Params(stat_filter=True)
    allp, L, P = O.instance_points("Car", lid, pse, p)
    cen = O.voxel_down_sample(pse[O.radius_outlier(pse, 1, 3.0)], 0.05)
    kept = cen[O.statistical_outlier(cen, 30, 0.3)]
    exp = kept[O.ball_query(kept, L, p.fuse_C)]
    assert np.array_equal(P, exp) and 0 < len(P) < len(cen) < len(pse)
