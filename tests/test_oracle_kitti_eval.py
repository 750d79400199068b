"""The AP-evaluator oracle (oracle/kitti_eval_oracle.py) against golden G9 -- the outputs of the reference's own
eval.py (tests/golden/capture_eval_golden.py) -- and against properties of the metric."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import eval_cases  # noqa: E402

from oracle import kitti_eval_oracle as KO  # noqa: E402

G9 = np.load(os.path.join(HERE, 'golden', 'g9_kitti_eval.npz'))
CASES = {tag: spec[3] for tag, spec in eval_cases.GOLDEN_CASES.items()}


def golden_annos(tag, side):
    pre = '%s/%s/' % (tag, side)
    return eval_cases.unpack({k[len(pre):]: G9[k] for k in G9.files if k.startswith(pre)})


def test_golden_inputs_are_what_the_generator_makes():
    for tag, (seed, frames, kw, _) in eval_cases.GOLDEN_CASES.items():
        gts, dts = eval_cases.make_annos(seed, frames, **kw)
        for side, annos in (('gt', gts), ('dt', dts)):
            for k, v in eval_cases.pack(annos).items():
                assert np.array_equal(G9['%s/%s/%s' % (tag, side, k)], v), (tag, side, k)


@pytest.mark.parametrize('tag', sorted(CASES))
@pytest.mark.parametrize('metric', [0, 1, 2])
def test_eval_class_reproduces_the_reference(tag, metric):
    gts, dts = golden_annos(tag, 'gt'), golden_annos(tag, 'dt')
    cls = [KO.CLASS_NAMES.index(c) for c in CASES[tag]]
    mo = KO.official_min_overlaps()[:, :, cls]
    aos = bool(len(dts) and any(len(d['alpha']) and d['alpha'][0] != -10 for d in dts)) and metric == 0
    r = KO.eval_class(gts, dts, cls, (0, 1, 2), metric, mo, aos)
    for key in ('recall', 'precision', 'orientation'):
        want = G9['%s/metric%d/%s' % (tag, metric, key)]
        assert r[key].shape == want.shape
        np.testing.assert_allclose(r[key], want, rtol=0, atol=1e-12, equal_nan=True, err_msg=key)
    assert np.nanmax(G9['%s/metric%d/precision' % (tag, metric)]) > 0.2          # the case is not degenerate


@pytest.mark.parametrize('tag', sorted(CASES))
def test_official_result_reproduces_the_reference(tag):
    gts, dts = golden_annos(tag, 'gt'), golden_annos(tag, 'dt')
    got = KO.official_result(gts, dts, CASES[tag])
    keys = [str(k) for k in G9['%s/official/keys' % tag]]
    assert sorted(got) == keys
    np.testing.assert_allclose([got[k] for k in keys], G9['%s/official/values' % tag], rtol=0, atol=1e-10, equal_nan=True)


def test_get_thresholds_reproduces_the_reference():
    sc = G9['thresholds/scores']
    for n_gt in (37, 500, 800):
        assert np.array_equal(np.array(KO.get_thresholds(sc, n_gt)), G9['thresholds/num_gt_%d' % n_gt])


def perfect_case(frames=30):
    """Every Car visible at all three levels, detections = the ground truth (enough objects for all 41 sample points)."""
    gts, _ = eval_cases.make_annos(3, frames, classes=('Car',), empty_frames=False)
    dts = []
    for g in gts:
        g['occluded'][:] = 0
        g['truncated'][:] = 0.0
        g['bbox'][:, 3] = g['bbox'][:, 1] + 90.0
        keep = np.array([n == 'Car' for n in g['name']], bool)
        d = {k: np.asarray(v)[keep].copy() for k, v in g.items()}
        d['score'] = np.linspace(0.9, 0.5, int(keep.sum()))
        dts.append(d)
    return gts, dts


def test_perfect_detections_score_100():
    gts, dts = perfect_case()
    assert sum(int((g['name'] == 'Car').sum()) for g in gts) >= 41
    res = KO.official_result(gts, dts, ['Car'])
    assert len(res) == 12
    for k, v in res.items():
        assert abs(v - 100.0) < 1e-9, (k, v)


def test_image_overlap_criteria():
    a = np.array([[0.0, 0.0, 10.0, 10.0]])
    b = np.array([[5.0, 5.0, 15.0, 15.0], [20.0, 20.0, 30.0, 30.0]])
    assert np.allclose(KO.image_box_overlap(a, b, -1), [[25.0 / 175.0, 0.0]])
    assert np.allclose(KO.image_box_overlap(a, b, 0), [[0.25, 0.0]])
    assert np.allclose(KO.image_box_overlap(a, b, 1), [[0.25, 0.0]])
