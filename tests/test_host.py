"""CPU-only tests of the host logic: calibration record, parameters, synthetic
scene generator (no GPU compute)."""
import os

import numpy as np
import pytest

from oracle import penet_oracle as O


def test_calibration_record_matches_oracle(golden_dir, tmp_path):
    from dfu3d_amd.calibration import Calibration
    g = np.load(golden_dir + "/g1_calib.npz")
    for tag in ("a", "b"):
        p = tmp_path / (tag + ".txt")
        p.write_bytes(bytes(g[tag + "_text"]))
        c = Calibration(str(p))
        o = O.Calibration(str(p))
        assert np.array_equal(c.P2, g[tag + "_P2"]) and np.array_equal(c.V2C, g[tag + "_V2C"])
        assert np.array_equal(c.M43, o.M43)                     # explicit FMA chain == C fmaf chain
        assert np.array_equal(c.M43, np.dot(c.V2C.T, c.R0.T))   # == the reference's np.dot here
        assert np.array_equal(c.Minv, o.Minv)
        r = c.record()
        assert r.shape == (48,) and r.dtype == np.float32
        assert np.array_equal(r[24:30], g[tag + "_scal"])


def test_round_f32_is_exact():
    from fractions import Fraction
    from dfu3d_amd.calibration import _round_f32
    rng = np.random.default_rng(3)
    for _ in range(2000):
        a, b, c = rng.normal(0, 10, 3).astype(np.float32)
        q = Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c))
        r = _round_f32(q)
        lo, hi = np.nextafter(r, np.float32(-np.inf)), np.nextafter(r, np.float32(np.inf))
        assert abs(Fraction(float(r)) - q) <= abs(Fraction(float(lo)) - q)
        assert abs(Fraction(float(r)) - q) <= abs(Fraction(float(hi)) - q)


def test_params_match_oracle_defaults():
    from dfu3d_amd.params import Params
    p, o = Params(), O.Params()
    for k in O.Params.__dataclass_fields__:
        assert getattr(p, k) == getattr(o, k), k
    assert p.thetas()[0] == 89
    assert p.instance_radii("Car") == (3.0, 3.0)
    assert p.instance_radii("Pedestrian") == (-1.0, 0.6)
    assert p.instance_radii("Person ride a bike") == (0.0, 0.0)


def test_synthetic_scene_shapes_and_oracle_runs():
    from dfu3d_amd import synth
    from dfu3d_amd.params import Params
    s = synth.make_scene(7, H=90, W=160, M=4, cams=2, dense=True, k_min=10, k_max=12)
    assert s.points.shape == (34720, 4) and s.depth.shape == (2, 90, 160)
    assert s.masks.shape == (2, 4, 90, 160)
    p = Params(bounds_hw=(90, 160), fov_hw=(90, 160))
    b = synth.to_view_batch([s], p, "cpu")
    assert b.calib.shape == (2, 48) and b.view_frame.tolist() == [0, 0]
    cal = s.calibs[0]
    oc = O.Calibration({"P2": cal.P2, "R0": cal.R0, "Tr_velo2cam": cal.V2C})
    lid, _ = O.fov_filter(s.points.numpy(), oc, p.fov_hw)
    n = int(s.n_inst[0])
    op = O.Params(bounds_hw=(90, 160), fov_hw=(90, 160))
    res = O.depth2pointsrgbpm(s.depth[0].numpy().copy()[:, :, None], None, oc, lid, O.NUSC_CLASSES,
                              s.masks[0][:n].numpy().astype(np.float32), s.inst_class[0][:n].numpy(),
                              s.inst_box[0][:n].numpy(), op, want_points=False)
    assert res.plane.shape == (4,)
    assert abs(res.plane[3] - 1.84) < 0.1       # ground at z = -1.84


def test_reader_pool_fills_shared_staging(tmp_path):
    """dfu3d_amd.reader_pool: forked workers read frames into shared memory at fixed strides; the bytes equal what
    pipeline.read_frame returns (CPU only: the pool refuses to start once the GPU is initialised)."""
    from multiprocessing import shared_memory
    import numpy as np
    from dfu3d_amd import kitti_io, synth
    from dfu3d_amd.params import NUSC_CLASSES
    from dfu3d_amd.pipeline import read_frame
    from dfu3d_amd.reader_pool import ReaderPool
    H, W, M = 48, 64, 4
    root = str(tmp_path / "k")
    for f in range(3):
        s = synth.make_scene(50 + f, H=H, W=W, M=M, cams=1, dense=True, k_min=5, k_max=7)
        n = int(s.n_inst[0])
        kitti_io.write_frame(root, f, s.points.numpy(), s.calibs[0], np.full((H, W, 3), 7 + f, np.uint8),
                             s.masks[0][:n].numpy(), s.inst_class[0][:n].numpy(), np.full(n, 0.9, np.float32),
                             s.inst_box[0][:n].numpy(), NUSC_CLASSES, s.depth[0].numpy(), compress=bool(f % 2))
    pool = ReaderPool(2)
    lay = ReaderPool.layout(4, H, W, 40000, 8, True)
    shm = shared_memory.SharedMemory(create=True, size=lay["bytes"])
    try:
        stems = ["%06d" % f for f in range(3)]
        metas = pool.read_batch(root, stems, None, None, True, shm.name, lay)
        for i, (st, m) in enumerate(zip(stems, metas)):
            ref = read_frame(root, st, want_image=True)
            assert m["stem"] == st and m["n"] == ref.points.shape[0] and m["m"] == ref.masks.shape[0]
            v = lambda off, shape, dt: np.ndarray(shape, dt, buffer=shm.buf, offset=off)
            assert np.array_equal(v(lay["pts"] + i * lay["cap_n"] * 16, (m["n"], 4), np.float32), ref.points)
            assert np.array_equal(v(lay["depth"] + i * H * W * 4, (H, W), np.float32), ref.depth)
            assert np.array_equal(v(lay["masks"] + i * 8 * H * W, (m["m"], H, W), np.uint8), ref.masks)
            assert np.array_equal(v(lay["image"] + i * H * W * 3, (H, W, 3), np.uint8), ref.image)
            assert np.array_equal(m["calib"], ref.calib.record()) and np.array_equal(m["classes"], ref.classes)
        import pytest
        with pytest.raises(RuntimeError):
            pool.read_batch(root, ["999999"], None, None, False, shm.name, lay)      # missing frame is reported
        assert len(pool.read_batch(root, stems[:1], None, None, False, shm.name, lay)) == 1   # and the pool lives on
    finally:
        pool.close()
        shm.close()
        shm.unlink()


def test_number_tables_of_the_prose_are_generated_from_the_committed_summaries():
    """Every block between `BEGIN GENERATED` / `END GENERATED` markers in DESIGN.md and profiles/README_r04.md is what
    tools/render_numbers.py writes from profiles/r04_* (kernel statistics, PMC counters, instruction mix, the bench line):
    a number typed by hand into one of them, or a profile refreshed without re-rendering, fails here."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "render_numbers.py"), "r04", "--check"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    text = open(os.path.join(root, "DESIGN.md")).read()
    assert text.count("BEGIN GENERATED") == text.count("END GENERATED") >= 5


def test_bench_parity_block_sees_extra_missing_and_moved_rows():
    """bench.py's `parity` block is the last line of defence of the timed configuration (in round 4 it was the only check that
    saw a race of large launches): equal rows -> rows_equal and within tolerance; one row more, one row less, another class
    or a box that moved by a centimetre -> not."""
    import numpy as np
    import torch
    sys_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    import sys
    sys.path.insert(0, sys_path)
    import bench
    rng = np.random.default_rng(3)
    exp, rows = [], []
    for f in range(2):
        for c in range(3):
            for j in range(2):
                v = rng.normal(0, 5, 12)
                exp.append((f, c, j, 0, 4, v))
                rows.append([100 + f, c, 4, j, 0] + list(v) + [0.5])          # gathered layout: global frame id first
    frame_ids = [100, 101]
    R = torch.tensor(np.array(rows, np.float64))
    ok = bench.parity_block(R, exp, 2, frame_ids)
    assert ok["rows_equal"] and ok["within_north_star_tolerance"] and ok["rows_gpu"] == 12
    extra = torch.cat((R, R[:1].clone() + torch.tensor([0, 0, 0, 0, 1] + [0.0] * 13, dtype=torch.float64)))   # one cluster more
    assert not bench.parity_block(extra, exp, 2, frame_ids)["rows_equal"]
    assert not bench.parity_block(R[1:], exp, 2, frame_ids)["rows_equal"]
    other = R.clone(); other[3, 2] = 7                                           # another class
    assert not bench.parity_block(other, exp, 2, frame_ids)["rows_equal"]
    moved = R.clone(); moved[5, 5 + 8] += 0.01                                   # x of a box by a centimetre
    b = bench.parity_block(moved, exp, 2, frame_ids)
    assert b["rows_equal"] and not b["within_north_star_tolerance"]
