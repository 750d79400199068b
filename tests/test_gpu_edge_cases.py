"""GPU edge cases the reference's data can produce: ragged / empty frames, views
without instances, empty depth, 32 instances per view, capacity overflows (flagged,
never silent), identical results for any chunk / stream layout."""
import numpy as np
import pytest
import torch

from oracle import penet_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _oracle_rows(scenes, p, dense):
    op = O.Params(**{k: getattr(p, k) for k in O.Params.__dataclass_fields__ if hasattr(p, k)})
    out, v = [], 0
    for s in scenes:
        pts = s.points.numpy()
        for c, cal in enumerate(s.calibs):
            oc = O.Calibration({"P2": cal.P2, "R0": cal.R0, "Tr_velo2cam": cal.V2C})
            lid, _ = O.fov_filter(pts, oc, p.fov_hw) if pts.shape[0] else (pts, None)
            n = int(s.n_inst[c])
            depth = s.depth[c].numpy().copy() if dense else np.zeros(tuple(s.depth[c].shape), np.float32)
            if lid.shape[0] >= 3:
                res = O.depth2pointsrgbpm(depth[:, :, None], None, oc, lid, O.NUSC_CLASSES,
                                          s.masks[c][:n].numpy().astype(np.float32), s.inst_class[c][:n].numpy(),
                                          s.inst_box[c][:n].numpy(), op, plane_key=v, want_points=False)
                out += [(v, r.inst, r.cluster, r) for r in res.rows]
            else:                       # no LiDAR: degenerate plane, pseudo points only
                res = O.depth2pointsrgbpm(depth[:, :, None], None, oc, np.zeros((0, 4), np.float32), O.NUSC_CLASSES,
                                          s.masks[c][:n].numpy().astype(np.float32), s.inst_class[c][:n].numpy(),
                                          s.inst_box[c][:n].numpy(), op, plane=np.array([0, 0, 1.0, 1e30]), want_points=False)
                out += [(v, r.inst, r.cluster, r) for r in res.rows]
            v += 1
    return out


def _compare(rows, exp, tol=1e-6):
    R = rows.cpu().numpy()
    assert R.shape[0] == len(exp), (R.shape[0], len(exp))
    for got, (v, j, k, r) in zip(R, exp):
        assert (int(got[0]), int(got[1]), int(got[2]), int(got[3])) == (v, j, k, r.cls)
        np.testing.assert_allclose(got[4:16], r.as_vector(), rtol=tol, atol=tol)


def test_ragged_and_empty_frames_and_views():
    _need_gpu()
    from dfu3d_amd import synth
    from dfu3d_amd.engine import PseudoBoxEngine
    from dfu3d_amd.params import Params
    H, W, M, cams = 180, 320, 5, 2
    p = Params(bounds_hw=(H, W), fov_hw=(H, W))
    scenes = [synth.make_scene(70 + f, H=H, W=W, M=M, cams=cams, dense=True, k_min=12, k_max=16,
                               rings=(32, 20, 8)[f], az=(1085, 700, 300)[f]) for f in range(3)]
    scenes[1].n_inst[0] = 0                       # a view without instances
    scenes[1].masks[1] = 0                        # a view whose masks are all zero
    scenes[2].depth[0] = 0                        # a view without depth
    scenes.append(synth.Scene(torch.zeros((0, 4)), scenes[0].calibs, scenes[0].depth.clone(), scenes[0].masks.clone(),
                              scenes[0].n_inst.clone(), scenes[0].inst_class, scenes[0].inst_box,
                              scenes[0].inst_score, scenes[0].boxes3d))          # a frame with no LiDAR points
    b = synth.to_view_batch(scenes, p, DEV, dense=True)
    cap_n = max(s.points.shape[0] for s in scenes)
    exp = _oracle_rows(scenes, p, True)
    ref = None
    for vpc, lanes in ((cams, 1), (2 * cams, 2), (4 * cams, 1), (1, 3)):
        eng = PseudoBoxEngine(p, H, W, M, cap_n, views_per_chunk=vpc, dense=True, cap_vox=1 << 16, lanes=lanes)
        rows, status = eng.run(b)
        assert status == 0
        _compare(rows, exp)
        if ref is None:
            ref = rows.clone()
        else:
            assert torch.equal(rows, ref)       # bit-identical for every chunk / stream layout


def test_thirty_two_instances_per_view():
    _need_gpu()
    from dfu3d_amd import synth
    from dfu3d_amd.engine import PseudoBoxEngine
    from dfu3d_amd.params import Params
    H, W, M, cams = 180, 320, 32, 2
    p = Params(bounds_hw=(H, W), fov_hw=(H, W))
    s = synth.make_scene(81, H=H, W=W, M=M, cams=cams, dense=True, k_min=60, k_max=70)
    assert int(s.n_inst.max()) > 8
    # make the last instance slot (bit 31) live as a copy of instance 0
    for c in range(cams):
        n = int(s.n_inst[c])
        if n < M:
            s.masks[c, M - 1] = s.masks[c, 0]
            s.inst_class[c, M - 1] = s.inst_class[c, 0]
            s.inst_box[c, M - 1] = s.inst_box[c, 0]
            s.inst_score[c, M - 1] = s.inst_score[c, 0]
            s.masks[c, n:M - 1] = 0
            s.n_inst[c] = M
    b = synth.to_view_batch([s], p, DEV, dense=True)
    eng = PseudoBoxEngine(p, H, W, M, s.points.shape[0], views_per_chunk=cams, dense=True, cap_vox=1 << 16)
    rows, status = eng.run(b)
    assert status == 0
    _compare(rows, _oracle_rows([s], p, True))
    assert (rows[:, 1] == M - 1).any()


def test_capacity_overflows_are_flagged_and_recoverable():
    _need_gpu()
    from dfu3d_amd import synth, stages as st
    from dfu3d_amd.engine import PseudoBoxEngine
    from dfu3d_amd.params import Params
    H, W, M, cams = 180, 320, 5, 2
    p = Params(bounds_hw=(H, W), fov_hw=(H, W))
    s = synth.make_scene(90, H=H, W=W, M=M, cams=cams, dense=True, k_min=14, k_max=18)
    b = synth.to_view_batch([s], p, DEV, dense=True)
    exp = _oracle_rows([s], p, True)
    n = s.points.shape[0]
    eng = PseudoBoxEngine(p, H, W, M, n, views_per_chunk=cams, dense=True, cap_vox=1 << 16, pool_per_view=64)
    _, status = eng.run(b)
    assert status & st.ST_POOL_OVERFLOW
    eng = PseudoBoxEngine(p, H, W, M, n, views_per_chunk=cams, dense=True, cap_vox=1 << 16, rows_per_view=1)
    _, status = eng.run(b)
    assert status & st.ST_ROW_OVERFLOW
    eng = PseudoBoxEngine(p, H, W, M, n, views_per_chunk=cams, dense=True, cap_vox=256)
    _, status = eng.run(b)
    assert status & st.ST_VOX_OVERFLOW
    # the engine re-initialises its bin table after a voxel overflow: a roomy twin agrees with the oracle
    eng2 = PseudoBoxEngine(p, H, W, M, n, views_per_chunk=cams, dense=True, cap_vox=1 << 16)
    eng2.table, eng2.lanes[0].table = eng.table, eng.table      # reuse the (re-initialised) table
    rows, status = eng2.run(b)
    assert status == 0
    _compare(rows, exp)


def test_chain_recovers_from_voxel_overflow():
    """ONE engine with chain=True (its bin table lives inside the chain workspace): a pass that overflows cap_vox
    leaves that table dirty; collect() re-initialises it, so the next pass of the SAME engine on a batch that fits is
    right again (a dirty table would give wrong voxels with status 0)."""
    _need_gpu()
    from dfu3d_amd import synth, stages as st
    from dfu3d_amd.engine import PseudoBoxEngine
    from dfu3d_amd.params import Params
    H, W, M, cams = 180, 320, 5, 2
    p = Params(bounds_hw=(H, W), fov_hw=(H, W))
    big = synth.make_scene(90, H=H, W=W, M=M, cams=cams, dense=True, k_min=14, k_max=18)
    small = synth.make_scene(91, H=H, W=W, M=M, cams=cams, dense=True, k_min=14, k_max=18)
    # the second batch keeps depth only in a window, so that it fits under the small cap
    d = small.depth.clone()
    keep = torch.zeros_like(d)
    keep[:, 60:120, 80:240] = d[:, 60:120, 80:240]
    small = synth.Scene(small.points, small.calibs, keep, small.masks, small.n_inst, small.inst_class, small.inst_box,
                        small.inst_score, small.boxes3d)
    n = max(big.points.shape[0], small.points.shape[0])
    exp = _oracle_rows([small], p, True)
    for chain in (True, False):
        eng = PseudoBoxEngine(p, H, W, M, n, views_per_chunk=cams, dense=True, cap_vox=4096, chain=chain)
        _, status = eng.run(synth.to_view_batch([big], p, DEV, dense=True))
        assert status & st.ST_VOX_OVERFLOW
        rows, status = eng.run(synth.to_view_batch([small], p, DEV, dense=True))
        assert status == 0
        _compare(rows, exp)


def test_wrapper_rejects_bad_operands():
    _need_gpu()
    from dfu3d_amd import stages as st
    from dfu3d_amd._lib import Dfu3dError
    z = torch.zeros(8, dtype=torch.float64, device=DEV)
    with pytest.raises(Dfu3dError):        # CPU tensor
        st.range_cluster(z.cpu(), z, torch.zeros(1, dtype=torch.int64, device=DEV),
                         torch.zeros(1, dtype=torch.int32, device=DEV), 1, 3.0, 0.001,
                         torch.zeros(8, dtype=torch.int32, device=DEV), 8)
    with pytest.raises(Dfu3dError):        # wrong dtype
        st.range_cluster(z.float(), z, torch.zeros(1, dtype=torch.int64, device=DEV),
                         torch.zeros(1, dtype=torch.int32, device=DEV), 1, 3.0, 0.001,
                         torch.zeros(8, dtype=torch.int32, device=DEV), 8)
    with pytest.raises(Dfu3dError):        # size mismatch
        st.range_cluster(z, z, torch.zeros(2, dtype=torch.int64, device=DEV),
                         torch.zeros(1, dtype=torch.int32, device=DEV), 1, 3.0, 0.001,
                         torch.zeros(8, dtype=torch.int32, device=DEV), 8)
