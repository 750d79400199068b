"""Golden G9: outputs of the reference's KITTI AP evaluator on seeded annotation sets.

Run in the build container (needs /root/reference; nothing at test time does):
    python tests/golden/capture_eval_golden.py        ->  tests/golden/g9_kitti_eval.npz

What is imported unmodified from /root/reference/pcdet/datasets/kitti/kitti_object_eval_python: eval.py (and, through
its relative import, rotate_iou.py), loaded as a two-file package so that pcdet/__init__ (compiled CUDA ops) is not
touched.  `numba` is absent from this image; like open3d / cv2 / spconv in capture_goldens.py it is provided as an
import-time stand-in whose `jit` / `cuda.jit` are the identity decorator, so eval.py's functions run as the plain
Python they are written in.  The one thing that cannot run is rotate_iou.py's numba-CUDA kernel: `rotate_iou_gpu_eval`
is replaced by oracle/iou3d_oracle.py's float64 polygon overlap rounded to float32 -- so G9 pins the evaluator's logic
(clean_data, image_box_overlap, d3_box_overlap_kernel, compute_statistics_jit, fused_compute_statistics,
get_thresholds, eval_class with its partitioning, get_mAP / _R40, get_official_eval_result), NOT the rotated-overlap leaf.
"""
import importlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF_EVAL = '/root/reference/pcdet/datasets/kitti/kitti_object_eval_python'
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

from oracle import kitti_eval_oracle as KO  # noqa: E402
import eval_cases  # noqa: E402


def _identity_jit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda fn: fn


numba = types.ModuleType('numba')
numba.jit = _identity_jit
numba.cuda = types.ModuleType('numba.cuda')
numba.cuda.jit = _identity_jit
sys.modules['numba'] = numba
sys.modules['numba.cuda'] = numba.cuda

pkg = types.ModuleType('ref_kitti_eval')
pkg.__path__ = [REF_EVAL]
sys.modules['ref_kitti_eval'] = pkg
ref_eval = importlib.import_module('ref_kitti_eval.eval')
ref_eval.rotate_iou_gpu_eval = lambda boxes, query, criterion=-1, device_id=0: KO.rotated_overlap_f32(boxes, query, criterion)

CASES = eval_cases.GOLDEN_CASES


def main():
    out = {}
    for tag, (seed, frames, kw, classes) in CASES.items():
        gts, dts = eval_cases.make_annos(seed, frames, **kw)
        for k, v in eval_cases.pack(gts).items():
            out['%s/gt/%s' % (tag, k)] = v
        for k, v in eval_cases.pack(dts).items():
            out['%s/dt/%s' % (tag, k)] = v
        cls_int = [KO.CLASS_NAMES.index(c) for c in classes]
        mo = KO.official_min_overlaps()[:, :, cls_int]
        compute_aos = kw.get('with_alpha', True)
        for metric in (0, 1, 2):
            with np.errstate(all='ignore'):
                r = ref_eval.eval_class(gts, dts, cls_int, (0, 1, 2), metric, mo, compute_aos and metric == 0)
            for key in ('recall', 'precision', 'orientation'):
                out['%s/metric%d/%s' % (tag, metric, key)] = r[key]
        with np.errstate(all='ignore'):
            text, ret = ref_eval.get_official_eval_result(gts, dts, classes)
        keys = sorted(ret)
        out['%s/official/keys' % tag] = np.array(keys)
        out['%s/official/values' % tag] = np.array([ret[k] for k in keys], np.float64)
        out['%s/official/text' % tag] = np.array(text)
        # get_range_eval_result (eval.py:916-1008): its own class table maps 'Car' / 'Pedestrian' to the numbers 0 / 1
        with np.errstate(all='ignore'):
            rtext, rret = ref_eval.get_range_eval_result(gts, dts, ['Car', 'Pedestrian'])
        rkeys = sorted(rret)
        out['%s/range/keys' % tag] = np.array(rkeys)
        out['%s/range/values' % tag] = np.array([rret[k] for k in rkeys], np.float64)
        out['%s/range/text' % tag] = np.array(rtext)
        # get_coco_eval_result (eval.py:836-894) cannot run in the reference: np.linspace is handed a float count (TypeError under
        # NumPy >= 1.18) and do_coco_style_eval unpacks four of do_eval's eight return values (ValueError)
        try:
            with np.errstate(all='ignore'):
                ref_eval.get_coco_eval_result(gts, dts, ['Car'])
            coco_raises = ''
        except (ValueError, TypeError) as e:
            coco_raises = '%s: %s' % (type(e).__name__, e)
        out['%s/coco/raises' % tag] = np.array(coco_raises)
        print(tag, 'frames', frames, 'official keys', len(keys), 'range keys', len(rkeys), 'coco:', coco_raises or 'ran')
    # get_thresholds on its own
    rng = np.random.default_rng(5)
    sc = np.round(rng.uniform(0, 1, 500), 3)
    out['thresholds/scores'] = sc
    for n_gt in (37, 500, 800):
        out['thresholds/num_gt_%d' % n_gt] = np.array(ref_eval.get_thresholds(sc.copy(), n_gt))
    np.savez_compressed(os.path.join(HERE, 'g9_kitti_eval.npz'), **out)
    print('wrote g9_kitti_eval.npz with', len(out), 'arrays')


if __name__ == '__main__':
    main()
