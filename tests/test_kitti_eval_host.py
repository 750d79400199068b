"""Host pieces of row f-3's evaluation step that need no GPU: threshold picking against golden G9, label readers, box
conversions."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))

from oracle import kitti_eval_oracle as KO  # noqa: E402


def test_get_thresholds_matches_the_reference_and_the_oracle():
    from dfu3d_amd.pcdet_kitti import eval as E
    G9 = np.load(os.path.join(HERE, 'golden', 'g9_kitti_eval.npz'))
    sc = G9['thresholds/scores']
    for n_gt in (37, 500, 800):
        assert np.array_equal(np.array(E.get_thresholds(sc, n_gt)), G9['thresholds/num_gt_%d' % n_gt])
    rng = np.random.default_rng(0)
    for _ in range(300):
        n = int(rng.integers(0, 300))
        s = np.round(rng.uniform(0, 1, n), int(rng.integers(1, 6)))
        g = int(rng.integers(max(n, 1), 3 * n + 2))
        assert E.get_thresholds(s, g) == KO.get_thresholds(s, g)
    assert E.get_thresholds([], 5) == []


def test_class_codes():
    from dfu3d_amd.pcdet_kitti import eval as E
    assert E._code('Car') == 0 and E._code('car') == 0 and E._code('Pedestrian') == 8
    assert E._code('Van') == 1000 and E._code('Person_sitting') == 1008
    assert E._code('DontCare') == -1 and E._code('Tram') == -1


def test_label_reader_and_result_rows(tmp_path):
    from dfu3d_amd.pcdet_kitti import kitti_common
    p = tmp_path / '000007.txt'
    p.write_text('Car -1 -1 0.5000 10.0000 20.0000 110.0000 90.0000 1.5000 1.9000 4.2000 1.0000 1.6000 20.0000 -1.5700 0.8700\n'
                 'Pedestrian 0.30 2 -0.1 1 2 3 4 1.8 0.6 0.7 -3 1.7 9 0.3 0.4\n')
    (tmp_path / 'notes.txt').write_text('ignored')
    a = kitti_common.get_label_annos(str(tmp_path))
    assert len(a) == 1
    a = a[0]
    assert a['name'].tolist() == ['Car', 'Pedestrian'] and a['occluded'].tolist() == [-1, 2]
    assert np.allclose(a['dimensions'][0], [4.2, 1.5, 1.9]) and np.allclose(a['location'][1], [-3, 1.7, 9])
    assert np.allclose(a['score'], [0.87, 0.4]) and a['bbox'].shape == (2, 4)
    (tmp_path / '000008.txt').write_text('')
    b = kitti_common.get_label_annos(str(tmp_path), [8])[0]
    assert b['bbox'].shape == (0, 4) and b['score'].shape == (0,) and b['dimensions'].shape == (0, 3)
    kept = kitti_common.filter_annos_low_score([a], 0.5)[0]
    assert kept['name'].tolist() == ['Car']


def test_box_conversions_round_trip(tmp_path):
    from dfu3d_amd import kitti_io, synth
    from dfu3d_amd.calibration import Calibration
    from dfu3d_amd.pcdet_kitti import box_utils, gt_database
    rng = np.random.default_rng(2)
    cal = synth.make_calibration(35.0, 900, 1600, rng)
    cpath = os.path.join(str(tmp_path), 'c.txt')
    kitti_io.write_calib(cpath, cal.P2, cal.R0, cal.V2C)
    calib = Calibration(cpath)
    boxes = np.zeros((5, 7), np.float32)
    boxes[:, 0] = rng.uniform(10, 40, 5)
    boxes[:, 1] = rng.uniform(-5, 5, 5)
    boxes[:, 2] = rng.uniform(-1.5, 0, 5)
    boxes[:, 3:6] = rng.uniform(0.5, 5, (5, 3))
    boxes[:, 6] = rng.uniform(-3, 3, 5)
    cam = box_utils.boxes3d_lidar_to_kitti_camera(boxes, calib)
    assert cam.shape == (5, 7) and np.allclose(cam[:, 3:6], boxes[:, [3, 5, 4]])
    assert np.allclose(cam[:, 6], -boxes[:, 6] - np.pi / 2)
    # back to LiDAR with the f-2 conversion (kitti_dataset.py:229-237)
    loc = gt_database.rect_to_lidar(calib, cam[:, 0:3])
    loc[:, 2] += cam[:, 4] / 2
    assert np.abs(loc - boxes[:, 0:3]).max() < 1e-4
    corners = box_utils.boxes3d_to_corners3d_kitti_camera(cam)
    assert corners.shape == (5, 8, 3) and corners.dtype == np.float32
    assert np.allclose(corners[:, 0:4, 1], cam[:, 1:2], atol=1e-5) and np.allclose(corners[:, 4:8, 1], cam[:, 1:2] - cam[:, 4:5], atol=1e-5)
    assert np.allclose(corners.mean(axis=1)[:, [0, 2]], cam[:, [0, 2]], atol=1e-4)
    d = np.linalg.norm(corners[:, 0] - corners[:, 2], axis=1)               # diagonal of the bottom face
    assert np.allclose(d, np.hypot(cam[:, 3], cam[:, 5]), atol=1e-4)
    img = box_utils.boxes3d_kitti_camera_to_imageboxes(cam, calib, image_shape=(900, 1600))
    assert img.shape == (5, 4) and (img[:, 0] <= img[:, 2]).all() and (img[:, 1] <= img[:, 3]).all()
    assert img.min() >= 0 and img[:, 2].max() <= 1599 and img[:, 3].max() <= 899


def _prediction_case(tmp_path, frames=30):
    import torch
    from dfu3d_amd import kitti_io, synth
    from dfu3d_amd.calibration import Calibration
    from dfu3d_amd.pcdet_kitti import gt_database
    rng = np.random.default_rng(4)
    cal = synth.make_calibration(35.0, 900, 1600, rng)
    cpath = os.path.join(str(tmp_path), 'calib.txt')
    kitti_io.write_calib(cpath, cal.P2, cal.R0, cal.V2C)
    calib = Calibration(cpath)
    batch = {'frame_id': ['%06d' % f for f in range(frames)], 'calib': [calib] * frames,
             'image_shape': [np.array([900, 1600])] * frames}
    preds = []
    for f in range(frames):
        n = 4
        boxes = np.zeros((n, 7), np.float32)
        depth = rng.uniform(12, 40, n)
        cam_loc = np.stack([depth * rng.uniform(-0.25, 0.25, n), rng.uniform(1.4, 1.8, n), depth], 1).astype(np.float32)
        boxes[:, 3:6] = np.array([4.2, 1.9, 1.6]) * rng.uniform(0.9, 1.1, (n, 3))
        boxes[:, 0:3] = gt_database.rect_to_lidar(calib, cam_loc)
        boxes[:, 2] += boxes[:, 5] / 2
        boxes[:, 6] = rng.uniform(-3, 3, n)
        preds.append({'pred_boxes': torch.from_numpy(boxes),
                      'pred_scores': torch.from_numpy(rng.uniform(0.3, 1, n).astype(np.float32)),
                      'pred_labels': torch.from_numpy(np.array([1, 1, 1, 2]))})
    return batch, preds


def test_predictions_to_result_files_and_back(tmp_path):
    """generate_prediction_dicts -> result files -> label readers; scored with the ORACLE evaluator here (the GPU
    evaluator does the same in tests/test_gpu_kitti_eval.py)."""
    from dfu3d_amd.pcdet_kitti import kitti_common, kitti_dataset
    batch, preds = _prediction_case(tmp_path)
    out = tmp_path / 'res'
    out.mkdir()
    annos = kitti_dataset.generate_prediction_dicts(batch, preds, ['Car', 'Pedestrian'], output_path=out)
    assert len(annos) == 30 and annos[0]['name'].tolist() == ['Car', 'Car', 'Car', 'Pedestrian']
    assert annos[3]['frame_id'] == '000003' and annos[0]['boxes_lidar'].shape == (4, 7)
    assert sorted(p.name for p in out.iterdir())[:2] == ['000000.txt', '000001.txt']
    first = (out / '000000.txt').read_text().splitlines()[0].split(' ')
    assert len(first) == 16 and first[0] == 'Car' and first[1:3] == ['-1', '-1']
    back = kitti_common.get_label_annos(str(out))
    for g, a in zip(back, annos):
        assert np.abs(g['location'] - a['location']).max() < 1e-4 and np.abs(g['dimensions'] - a['dimensions']).max() < 1e-4
        assert np.abs(g['bbox'] - a['bbox']).max() < 1e-3 and np.abs(g['score'] - a['score']).max() < 1e-4
        assert np.abs(g['rotation_y'] - a['rotation_y']).max() < 1e-4 and np.abs(g['alpha'] - a['alpha']).max() < 1e-4
        g['occluded'][:] = 0
        g['truncated'][:] = 0.0
    res = KO.official_result(back, annos, ['Car'])
    assert len(res) == 12 and all(abs(v - 100.0) < 1e-6 for v in res.values()), res
    # an empty prediction keeps the template's shapes and writes an empty file
    import torch
    empty = {'pred_boxes': torch.zeros((0, 7)), 'pred_scores': torch.zeros(0), 'pred_labels': torch.zeros(0, dtype=torch.long)}
    e = kitti_dataset.generate_prediction_dicts({'frame_id': ['000099'], 'calib': batch['calib'][:1],
                                                 'image_shape': batch['image_shape'][:1]}, [empty], ['Car'], output_path=out)
    assert e[0]['bbox'].shape == (0, 4) and (out / '000099.txt').read_text() == ''


def test_evaluate_entry_point(tmp_path, monkeypatch):
    """The standalone entry point reads the split's files and hands them to the evaluator (stubbed here: no GPU)."""
    from dfu3d_amd.pcdet_kitti import evaluate as ev
    lab, res = tmp_path / 'label_2', tmp_path / 'res'
    lab.mkdir(), res.mkdir()
    row = 'Car 0.0 0 0.1 10 20 110 90 1.5 1.9 4.2 1.0 1.6 20.0 -1.57'
    for i in (3, 5):
        (lab / ('%06d.txt' % i)).write_text(row + '\n')
        (res / ('%06d.txt' % i)).write_text(row + ' 0.9\n' + row + ' 0.05\n')
    (res / '000004.txt').write_text(row + ' 0.9\n')                   # not in the split: must not be read
    (tmp_path / 'val.txt').write_text('3\n5\n')
    seen = {}

    def fake(gt, dt, classes, PR_detail_dict=None):
        seen.update(gt=gt, dt=dt, classes=classes)
        return 'report', {}
    monkeypatch.setattr(ev.kitti_eval, 'get_official_eval_result', fake)
    assert ev.main(['--label_path', str(lab), '--result_path', str(res), '--label_split_file', str(tmp_path / 'val.txt'),
                    '--current_class', 'Car', '8', '--score_thresh', '0.1']) == 0
    assert len(seen['gt']) == 2 and len(seen['dt']) == 2 and seen['classes'] == ['Car', 8]
    assert [len(d['name']) for d in seen['dt']] == [1, 1] and seen['gt'][0]['score'].tolist() == [0.0]
    import pytest
    with pytest.raises(NotImplementedError):
        ev.evaluate(str(lab), str(res), str(tmp_path / 'val.txt'), coco=True)
