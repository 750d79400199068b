"""GPU tests of the radius filter beyond the stage tests of test_gpu_stages.py:
the joint pass over LiDAR + pseudo lists followed by dfu3d_ballquery_fuse_joint against the oracle's
filter-filter-fuse sequence (my_loader.py:581-605), and the keep mask of a BENCH frame's pool -- 50 000-point
segments, the long-query and overflow-list paths -- against a brute-force float64 count, for nb_points 1 and 4."""
import numpy as np
import pytest
import torch

from oracle import penet_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def st():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from dfu3d_amd import stages
    return stages


def _t(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV).contiguous()


def test_joint_filter_then_joint_fuse_equals_filter_filter_fuse(st):
    """ONE radius-filter pass over the 2S lists (A = LiDAR, B = pseudo; no compaction) + dfu3d_ballquery_fuse_joint
    == remove_radius_outlier(A), remove_radius_outlier(B), BallQuery, cat -- including instances whose LiDAR list the
    filter empties (isolated points, or radius -1: hazard H4), where the fuse is skipped and every filtered pseudo point
    stays (my_loader.py:602), LiDAR lists beyond both hash-table builds (brute-force tiles) and empty lists."""
    rng = np.random.default_rng(2024)
    #        na    nb   r_a   r_b
    cases = [(0, 40, 3.0, 0.6), (30, 0, 3.0, 3.0), (60, 700, 3.0, 0.6), (900, 4000, 3.0, 3.0), (5, 5, 0.6, 0.6),
             (4200, 3000, 3.0, 3.0), (7, 300, 3.0, 0.6), (12, 500, -1.0, 0.6), (1500, 2500, 0.6, 0.6), (300, 600, 0.0, 3.0)]
    segsA, segsB = [], []
    for k, (na, nb, ra, rb) in enumerate(cases):
        a = rng.normal(0, 2.0, (na, 3))
        b = rng.normal(0, 2.0, (nb, 3))
        if na and nb:
            h = nb // 2
            b[:h] = a[rng.integers(0, na, h)] + rng.normal(0, 0.06, (h, 3))
            far = rng.random(nb) < 0.2                              # isolated pseudo points: the filter drops them
            b[far] = rng.uniform(-300, 300, (int(far.sum()), 3))
            fa = rng.random(na) < 0.15                              # and isolated LiDAR points
            a[fa] = rng.uniform(-300, 300, (int(fa.sum()), 3)) + 1000.0
        if k == 6:                                                  # every LiDAR point isolated: the filter empties the list
            a = rng.uniform(-1, 1, (na, 3)) + np.arange(na)[:, None] * 50.0 + 2000.0
        segsA.append(a)
        segsB.append(b)
    S = len(cases)
    chunks, base_a, base_b, cur = [], [], [], 0
    for a, b in zip(segsA, segsB):
        base_a.append(cur); cur += len(a)
        base_b.append(cur); cur += len(b)
        chunks += [a, b]
    P = np.concatenate(chunks)
    cap = len(P) + 16
    Pp = np.full((cap, 3), 9.0); Pp[:len(P)] = P
    px, py, pz = _t(Pp[:, 0]), _t(Pp[:, 1]), _t(Pp[:, 2])
    cnt_a = _t(np.array([len(a) for a in segsA], np.int32))
    cnt_b = _t(np.array([len(b) for b in segsB], np.int32))
    ta, tb = _t(np.array(base_a, np.int64)), _t(np.array(base_b, np.int64))
    ra = np.array([c[2] for c in cases]); rb = np.array([c[3] for c in cases])
    # the joint table, as dfu3d_segments_build lays it out: s < S the A lists, S + s the B lists
    base_ab = torch.cat([ta, tb]); cnt_ab = torch.cat([cnt_a, cnt_b]); rad_ab = _t(np.concatenate([ra, rb]))
    tile_off = torch.zeros(2 * S + 2, dtype=torch.int32, device=DEV)
    flags = torch.zeros(cap, dtype=torch.uint8, device=DEV)
    queue = torch.zeros(st.rf_queue_ints(cap), dtype=torch.int32, device=DEV)
    st.radius_filter(px, py, pz, base_ab, cnt_ab, rad_ab, 1, 2 * S, cap, tile_off, flags, queue,
                     phases=st.RF_ALL & ~st.RF_COMPACT)
    st.ballquery_fuse(px, py, pz, ta, cnt_a, tb, cnt_b, 0.1, S, cap, tile_off, flags, masked="joint")
    torch.cuda.synchronize()
    X = torch.stack([px, py, pz], 1).cpu().numpy()
    nca, ncb, nbb = cnt_a.cpu().numpy(), cnt_b.cpu().numpy(), tb.cpu().numpy()

    def filt(pts, r):
        if len(pts) == 0 or r == 0.0:
            return pts
        if r < 0.0:
            return pts[:0]
        return pts[O.radius_outlier(pts, 1, r)]

    skipped = 0
    for s, (a, b) in enumerate(zip(segsA, segsB)):
        a1, b1 = filt(a, ra[s]), filt(b, rb[s])
        if len(a1) and len(b1):
            keep = O.ball_query(b1, a1, 0.1)
        else:
            keep = np.ones(len(b1), bool)
            skipped += int(len(a) > 0 and len(b1) > 0)
        exp = np.concatenate([a1, b1[keep]])
        assert nca[s] == len(a1), (s, nca[s], len(a1))
        assert ncb[s] == int(keep.sum()), (s, ncb[s], int(keep.sum()))
        assert nbb[s] == base_a[s] + len(a1), s
        assert np.array_equal(X[base_a[s]:base_a[s] + len(exp)], exp), s
    assert skipped >= 2                      # the emptied-list cases were really there


def _brute_force_keep(P, r, nb):
    """remove_radius_outlier's rule on the GPU in float64: keep iff #(d2 < r2, self included) > nb."""
    n = P.shape[0]
    if r == 0.0:
        return torch.ones(n, dtype=torch.bool, device=P.device)
    if not r > 0.0:
        return torch.zeros(n, dtype=torch.bool, device=P.device)
    out = torch.empty(n, dtype=torch.bool, device=P.device)
    for q0 in range(0, n, 4096):
        Q = P[q0:q0 + 4096]
        c = torch.zeros(Q.shape[0], dtype=torch.int64, device=P.device)
        for j0 in range(0, n, 16384):
            R = P[j0:j0 + 16384]
            dx = Q[:, None, 0] - R[None, :, 0]; dy = Q[:, None, 1] - R[None, :, 1]; dz = Q[:, None, 2] - R[None, :, 2]
            d = dx * dx; d = d + dy * dy; d = d + dz * dz
            c += (d < r * r).sum(1)
        out[q0:q0 + 4096] = c > nb
    return out


@pytest.mark.parametrize("nb", [1, 4])
def test_keep_mask_of_a_bench_frame_equals_the_brute_force_count(st, nb):
    """The pool the engine hands the filter for ONE frame of bench.py's workload (34 720-point sweep, 6 x 1600x900,
    dense: ~180 000 points, pseudo lists of up to ~50 000 points next to LiDAR lists of a few hundred) through
    FLAGS | RESOLVE, every flag against a float64 recount of EVERY segment.  nb = 4 drives k_rf_stream<false>, the
    per-range U slots filling up, the segment overflow lists and k_rf_ranges on the long segments."""
    from dfu3d_amd import synth
    from dfu3d_amd.engine import PseudoBoxEngine
    from dfu3d_amd.params import Params
    p = Params()
    sc = synth.make_scene(5, dense=True, device=DEV, k_min=30, k_max=40)
    b = synth.to_view_batch([sc], p, DEV)
    eng = PseudoBoxEngine(p, 900, 1600, 8, 34720, views_per_chunk=6)
    snap = {}
    orig = st.radius_filter

    def hook(px, py, pz, base, cnt, radius, nb_, S, pool_cap, tile_off, flags, queue, phases=st.RF_ALL, shadow=None, n_used=None):
        if not snap:
            snap.update(px=px.clone(), py=py.clone(), pz=pz.clone(), base=base.clone(), cnt=cnt.clone(), radius=radius.clone(),
                        S=S, shadow=shadow.clone(), n_used=n_used.clone())
        return orig(px, py, pz, base, cnt, radius, nb_, S, pool_cap, tile_off, flags, queue, phases=phases, shadow=shadow,
                    n_used=n_used)

    st.radius_filter = hook
    try:
        eng.run(b)
    finally:
        st.radius_filter = orig
    torch.cuda.synchronize()
    assert snap and int(snap["cnt"].max()) > 20000                   # a long pseudo list is in there
    eng.flags.fill_(7)                                               # every flag of a listed position must be WRITTEN
    orig(snap["px"], snap["py"], snap["pz"], snap["base"], snap["cnt"], snap["radius"], nb, snap["S"], eng.pool_cap,
         eng.tile_off, eng.flags, eng.queue, phases=st.RF_FLAGS | st.RF_RESOLVE, shadow=snap["shadow"], n_used=snap["n_used"])
    torch.cuda.synchronize()
    flags = eng.flags
    base = snap["base"].cpu().numpy(); cnt = snap["cnt"].cpu().numpy(); rad = snap["radius"].cpu().numpy()
    X = torch.stack([snap["px"], snap["py"], snap["pz"]], 1)
    bad = tot = dropped = 0
    for s in np.nonzero(cnt > 0)[0]:
        n = int(cnt[s])
        exp = _brute_force_keep(X[base[s]:base[s] + n], float(rad[s]), nb)
        got = flags[base[s]:base[s] + n]
        assert int((got > 1).sum()) == 0, ("unwritten flags in segment", int(s))
        bad += int(((got != 0) != exp).sum()); tot += n; dropped += int((~exp).sum())
    assert tot > 100000 and dropped > 100
    assert bad == 0, "%d of %d flags differ from the brute-force count" % (bad, tot)


# ------------------------------------------------------------------ a11: voxel down-sample + statistical filter
def test_voxel_down_sample_matches_oracle(st):
    """dfu3d_voxel_down_sample against oracle.voxel_down_sample (Open3D VoxelDownSample restated, my_loader0.py:734):
    the centroids EQUAL bit for bit -- every sum is formed in list order on both sides -- and in the defined order
    (first-seen voxel).  Sizes around the 1024-point chunk of the kernel, voxels that span several chunks, duplicates,
    a disabled segment, an empty one."""
    rng = np.random.default_rng(31)
    segs = []
    for n in (0, 1, 2, 37, 1023, 1024, 1025, 3000, 7000):
        p_ = np.cumsum(rng.normal(0, 0.015, (n, 3)), 0) + rng.uniform(-30, 30, 3)     # a surface-like run: 1-6 points per voxel
        if n > 100:
            back = rng.integers(0, n, n // 10)                                        # revisits of earlier voxels, chunks later
            p_[back] = p_[rng.integers(0, n, n // 10)] + rng.normal(0, 0.004, (n // 10, 3))
        segs.append(p_)
    segs.append(np.repeat(rng.normal(0, 1, (1, 3)), 2500, 0))                         # one voxel, 2 500 points, three chunks
    segs.append(rng.uniform(-2, 2, (4000, 3)))                                        # nearly every point its own voxel
    segs.append(rng.normal(0, 1, (300, 3)))                                           # disabled: must stay as it is
    S = len(segs)
    enable = np.ones(S, np.int32); enable[-1] = 0
    base, cur, chunks = [], 5, [np.full((5, 3), 4242.0)]
    for p_ in segs:
        base.append(cur); chunks += [p_, np.full((3, 3), 777.0)]; cur += len(p_) + 3
    P = np.concatenate(chunks)
    cap = len(P) + 8
    Pp = np.full((cap, 3), 555.0); Pp[:len(P)] = P
    px, py, pz = _t(Pp[:, 0]), _t(Pp[:, 1]), _t(Pp[:, 2])
    cnt = _t(np.array([len(p_) for p_ in segs], np.int32))
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    scratch = torch.empty(st.voxel_down_sample_scratch_bytes(cap), dtype=torch.uint8, device=DEV)
    outs = []
    for rep in range(2):                                                              # deterministic: two runs, the same bits
        for t_, col in ((px, 0), (py, 1), (pz, 2)):
            t_.copy_(_t(Pp[:, col]))
        cnt.copy_(_t(np.array([len(p_) for p_ in segs], np.int32)))
        scratch.fill_(0xAB if rep else 0)                                             # the scratch needs no initialisation
        st.voxel_down_sample(px, py, pz, _t(np.array(base, np.int64)), cnt, _t(enable), 0.05, S, cap, scratch, status)
        torch.cuda.synchronize()
        outs.append((cnt.cpu().numpy().copy(), torch.stack([px, py, pz], 1).cpu().numpy().copy()))
    assert int(status.item()) == 0
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    n_out, X = outs[0]
    for s, p_ in enumerate(segs):
        exp = O.voxel_down_sample(p_, 0.05) if enable[s] else p_
        assert n_out[s] == len(exp), (s, n_out[s], len(exp))
        assert np.array_equal(X[base[s]:base[s] + len(exp)], exp), s
    assert np.all(X[:5] == 4242.0)                                                    # nothing outside the segments was touched


def test_engine_with_the_statistical_pair_matches_oracle_at_bench_size():
    """The path with the reference's dormant pair switched on (my_loader0.py:734-735: voxel_down_sample(0.05) ->
    remove_statistical_outlier(30, 0.3) on the pseudo points of every instance) on one frame of bench.py's workload -- 34 720-point
    sweep, 6 views of 1600x900, dense: pseudo lists of tens of thousands of points go through the voxel hash and the k-NN
    kernel.  Every label row against the oracle; the chain (one C call) gives the same bits."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from dfu3d_amd import synth
    from dfu3d_amd.engine import PseudoBoxEngine
    from dfu3d_amd.params import Params
    H, W, M, cams = 900, 1600, 8, 6
    p = Params(stat_filter=True)
    scenes = [synth.make_scene(21, H=H, W=W, M=M, cams=cams, dense=True, k_min=30, k_max=40)]
    b = synth.to_view_batch(scenes, p, DEV, dense=True)
    cap_n = scenes[0].points.shape[0]
    eng = PseudoBoxEngine(p, H, W, M, cap_n, views_per_chunk=cams, dense=True, cap_vox=1 << 18, pool_per_view=1 << 17)
    rows, status = eng.run(b)
    assert status == 0
    op = O.Params(stat_filter=True)
    exp = []
    s = scenes[0]
    for c, cal in enumerate(s.calibs):
        oc = O.Calibration({"P2": cal.P2, "R0": cal.R0, "Tr_velo2cam": cal.V2C})
        lid, _ = O.fov_filter(s.points.numpy(), oc, p.fov_hw)
        n = int(s.n_inst[c])
        res = O.depth2pointsrgbpm(s.depth[c].numpy().copy()[:, :, None], None, oc, lid, O.NUSC_CLASSES,
                                  s.masks[c][:n].numpy().astype(np.float32), s.inst_class[c][:n].numpy(),
                                  s.inst_box[c][:n].numpy(), op, plane_key=c, want_points=False)
        exp += [(c, r.inst, r.cluster, r) for r in res.rows]
    R = rows.cpu().numpy()
    assert R.shape[0] == len(exp) and len(exp) >= 5, (R.shape[0], len(exp))
    for got, (v, j, k, r) in zip(R, exp):
        assert (int(got[0]), int(got[1]), int(got[2]), int(got[3])) == (v, j, k, r.cls)
        np.testing.assert_allclose(got[4:16], r.as_vector(), rtol=1e-6, atol=1e-6)
    rows2, st2 = PseudoBoxEngine(p, H, W, M, cap_n, views_per_chunk=cams, dense=True, cap_vox=1 << 18,
                                 pool_per_view=1 << 17, chain=True).run(b)
    assert st2 == 0 and torch.equal(rows, rows2)
