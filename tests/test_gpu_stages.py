"""GPU parity tests: every C-ABI stage of libdfu3d_hip.so against the CPU oracle
on the same seeded inputs (bit-exact for index/membership results, 1e-9 for
fp64 geometry).  Run with `-m gpu` on an MI355X."""
import math

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


def _t(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV).contiguous()


def _rand_calib(rng, yaw_deg=0.0, H=900, W=1600):
    from dfu3d_amd import synth
    return synth.make_calibration(yaw_deg, H, W, rng)


def _oracle_calib(c):
    return O.Calibration({"P2": c.P2, "R0": c.R0, "Tr_velo2cam": c.V2C})


# ------------------------------------------------------------------ segments helper
def _pool_from_segments(segs, pad=3):
    """segs: list of (n_i,3) fp64 arrays -> pool tensors + base/cnt (with gaps)."""
    base, cnt, chunks, cur = [], [], [], 0
    for s in segs:
        base.append(cur)
        cnt.append(len(s))
        chunks.append(np.asarray(s, np.float64).reshape(-1, 3))
        chunks.append(np.full((pad, 3), 777.0))
        cur += len(s) + pad
    allp = np.concatenate(chunks) if chunks else np.zeros((0, 3))
    cap = max(len(allp), 1) + 8
    P = np.full((cap, 3), 555.0)
    P[:len(allp)] = allp
    return P, np.array(base, np.int64), np.array(cnt, np.int32), cap


def _clustered_points(rng, n, spread=1.0, outliers=0.05):
    k = max(1, n // 40)
    centers = rng.uniform(-20, 20, (k, 3)) * np.array([1, 1, 0.1])
    p = centers[rng.integers(0, k, n)] + rng.normal(0, spread, (n, 3))
    m = rng.random(n) < outliers
    p[m] = rng.uniform(-80, 80, (int(m.sum()), 3))
    return p


def test_float32_tier_never_disagrees_with_fp64(st):
    """Tier 1 of the bin classification (float32 back-projection, edge tables in cos / tan space) takes a decision only
    outside its error bounds.  Run it against the fp64 classification on 4e7 random pixels per case (six synthetic
    cameras, product geometry and two odd ones, both key axes, depths from centimetres to 95 m): not one decided pixel
    may differ in bin or voxel key, and tier 1 must decide nearly all of them."""
    from dfu3d_amd import synth
    rng = np.random.default_rng(0)
    geoms = [st.make_geom(),                                                       # the product's la_sampling grid
             st.make_geom(theta_min=0.3, z_max=3.0, vsize=(200.0, 0.0137, 0.0071), vgrid=(1, 800, 1500)),
             st.make_geom(theta_min=1.2, z_max=0.5, vsize=(200.0, 0.0005, 0.0005), vgrid=(1, 20000, 20000))]
    total = 0
    for gi, (geom, _) in enumerate(geoms):
        for H, W in ((900, 1600), (180, 320)):
            for yaw in synth.CAM_YAWS_DEG[:: (1 if gi == 0 else 3)]:
                cal = synth.make_calibration(yaw, H, W, rng)
                for key_axis, (d_lo, d_hi) in ((1, (0.01, 3.0)), (2, (0.5, 95.0))):
                    r = st.selftest_classify(cal.record(), H, W, geom, key_axis, 4_000_000, seed=int(yaw) + gi,
                                             d_lo=d_lo, d_hi=d_hi)
                    assert r["tried"] == 4_000_000 and r["wrong"] == 0, (gi, H, W, yaw, key_axis, r)
                    if gi == 0 and d_hi > 90:
                        assert r["undecided"] < 0.03 * r["tried"] and r["kept"] > 0, (H, W, yaw, r)
                    total += r["kept"]
    assert total > 10_000_000


def test_fp32_backprojection_bound_holds_on_this_device(st):
    """Tier 1 of the bin classification starts from a float32 back-projection (nine FMAs per pixel) and trusts it only
    outside a bound on its error: measure |float32 - fp64| / bound over 3e7 pixels per camera (six synthetic cameras,
    full 900x1600 and a small image, depths from centimetres to 120 m) -- it must stay under one half."""
    from dfu3d_amd import synth
    rng = np.random.default_rng(0)
    for H, W in ((900, 1600), (180, 320)):
        for yaw in synth.CAM_YAWS_DEG:
            cal = synth.make_calibration(yaw, H, W, rng)
            for d_lo, d_hi in ((0.01, 3.0), (0.5, 120.0)):
                q = st.selftest_backproject(cal.record(), H, W, 5_000_000, seed=int(yaw) + 7, d_lo=d_lo, d_hi=d_hi)
                assert 0.0 < q < 0.5, (H, W, yaw, d_lo, q)


# ------------------------------------------------------------------ a9
def test_segments_build_orders_every_list_across_chunks_and_waves(st):
    """Per-view bit sets -> per-instance ordered lists (my_loader.py:547-565 semantics: an item belongs to every instance
    whose bit it carries, lists keep the item order, LiDAR list directly followed by the pseudo list).  Item counts sit
    on the kernel's granularities (64-item ballots, 256-item wave ranges, 2048-item steps, 32768-item chunks) and next
    to them; the float32 shadow and the joint 2S table are checked too."""
    rng = np.random.default_rng(33)
    M = 6
    a_ns = [0, 1, 63, 64, 65, 255, 2049, 5000]
    b_ns = [70001, 32768, 32769, 0, 2048, 65536, 40000, 1]
    V = len(a_ns)
    a_cap, b_cap = 5120, 70016
    def items(ns, cap):
        bits = np.zeros((V, cap), np.int32)
        xyz = rng.normal(0, 30, (V, cap, 3))
        for v, n in enumerate(ns):
            b = np.zeros(n, np.int64)
            for j in range(M):
                b |= (rng.random(n) < (0.02 if j else 0.6)).astype(np.int64) << j
            b[rng.random(n) < 0.3] = 0
            bits[v, :n] = b.astype(np.int32)
            bits[v, n:] = 0x3F                                     # stale words beyond the count must be ignored
        return bits, xyz
    abits, axyz = items(a_ns, a_cap)
    bbits, bxyz = items(b_ns, b_cap)
    S = V * M
    total = sum(int(((abits[v, :a_ns[v]] >> j) & 1).sum() + ((bbits[v, :b_ns[v]] >> j) & 1).sum()) for v in range(V) for j in range(M))
    cap = total + 64
    dev = DEV
    t = lambda a, dt=None: torch.as_tensor(np.ascontiguousarray(a)).to(dev) if dt is None else torch.as_tensor(np.ascontiguousarray(a)).to(dev).to(dt)
    px, py, pz = (torch.full((cap,), 7.0, dtype=torch.float64, device=dev) for _ in range(3))
    base_a, base_b = torch.zeros(S, dtype=torch.int64, device=dev), torch.zeros(S, dtype=torch.int64, device=dev)
    cnt_a, cnt_b = torch.zeros(S, dtype=torch.int32, device=dev), torch.zeros(S, dtype=torch.int32, device=dev)
    cursor = torch.zeros(1, dtype=torch.int64, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    rad_a = t(rng.uniform(0.3, 3.0, S)); rad_b = t(rng.uniform(0.3, 3.0, S))
    shadow = torch.zeros(st.shadow_floats(cap), dtype=torch.float32, device=dev)
    base_ab = torch.zeros(2 * S, dtype=torch.int64, device=dev)
    cnt_ab = torch.zeros(2 * S, dtype=torch.int32, device=dev)
    rad_ab = torch.zeros(2 * S, dtype=torch.float64, device=dev)
    st.segments_build(t(abits.reshape(-1)), t(axyz[..., 0].reshape(-1)), t(axyz[..., 1].reshape(-1)), t(axyz[..., 2].reshape(-1)),
                      t(np.array(a_ns, np.int32)), a_cap,
                      t(bbits.reshape(-1)), t(bxyz[..., 0].reshape(-1)), t(bxyz[..., 1].reshape(-1)), t(bxyz[..., 2].reshape(-1)),
                      t(np.array(b_ns, np.int32)), b_cap, V, M, cap, cursor, px, py, pz, base_a, cnt_a, base_b, cnt_b, status,
                      rad_a=rad_a, rad_b=rad_b, shadow=shadow, base_ab=base_ab, cnt_ab=cnt_ab, rad_ab=rad_ab)
    torch.cuda.synchronize()
    assert int(status.item()) == 0 and int(cursor.item()) == total
    P = torch.stack([px, py, pz], 1).cpu().numpy()
    sh = shadow[:4 * cap].view(-1, 4).cpu().numpy()
    ba, ca, bb, cb = base_a.cpu().numpy(), cnt_a.cpu().numpy(), base_b.cpu().numpy(), cnt_b.cpu().numpy()
    assert np.array_equal(base_ab.cpu().numpy(), np.concatenate([ba, bb])) and np.array_equal(cnt_ab.cpu().numpy(), np.concatenate([ca, cb]))
    run = 0
    for v in range(V):
        for j in range(M):
            s = v * M + j
            ea = axyz[v, :a_ns[v]][((abits[v, :a_ns[v]] >> j) & 1) == 1]
            eb = bxyz[v, :b_ns[v]][((bbits[v, :b_ns[v]] >> j) & 1) == 1]
            assert (ca[s], cb[s]) == (len(ea), len(eb)), (v, j)
            assert ba[s] == run and bb[s] == run + len(ea)          # dense pool, A directly followed by B
            assert np.array_equal(P[ba[s]:ba[s] + ca[s]], ea) and np.array_equal(P[bb[s]:bb[s] + cb[s]], eb), (v, j)
            assert np.array_equal(sh[bb[s]:bb[s] + cb[s], :3], eb.astype(np.float32))
            seg_word = sh[bb[s]:bb[s] + cb[s], 3].view(np.uint32) >> 16
            assert (seg_word == S + s).all()
            run += len(ea) + len(eb)


# ------------------------------------------------------------------ a10
def test_radius_filter_matches_oracle(st):
    rng = np.random.default_rng(10)
    sizes = [0, 1, 2, 5, 63, 64, 65, 255, 256, 257, 700, 1024, 1025, 3000, 5200]
    segs = [_clustered_points(rng, n) for n in sizes]
    segs.append(np.repeat(rng.normal(0, 1, (1, 3)), 9, 0))          # duplicates
    radius = np.array([3.0, 0.6] * 8)[:len(segs)]
    radius[3] = -1.0                                                   # H4: drop all
    radius[4] = 0.0                                                    # H5: keep all
    for nb in (1, 4):
        P, base, cnt, cap = _pool_from_segments(segs)
        px, py, pz = _t(P[:, 0]), _t(P[:, 1]), _t(P[:, 2])
        S = len(segs)
        seg_cnt = _t(cnt)
        st.radius_filter(px, py, pz, _t(base), seg_cnt, _t(radius), nb, S, cap,
                         torch.zeros(S + 1, dtype=torch.int32, device=DEV),
                         torch.zeros(cap, dtype=torch.uint8, device=DEV),
                         torch.zeros(st.rf_queue_ints(cap), dtype=torch.int32, device=DEV))
        torch.cuda.synchronize()
        out_cnt = seg_cnt.cpu().numpy()
        X = torch.stack([px, py, pz], 1).cpu().numpy()
        for s, pts in enumerate(segs):
            if radius[s] < 0:
                keep = np.zeros((0,), np.int64)
            elif radius[s] == 0:
                keep = np.arange(len(pts))
            else:
                keep = O.radius_outlier(pts, nb, radius[s])
            assert out_cnt[s] == len(keep), (s, nb, out_cnt[s], len(keep))
            got = X[base[s]:base[s] + out_cnt[s]]
            assert np.array_equal(got, pts[keep]), (s, nb)


def test_radius_filter_pairs_at_the_threshold(st):
    """The filter screens in float32 and must hand every comparison that float32 cannot decide to the fp64
    predicate d2 < r2 (strict, nanoflann): pairs at distance r*(1 +- 1e-5 ... 1e-15), exactly r, far from the
    origin (large coordinates = large float32 rounding) and with fp64 coordinates that are not float32 values."""
    rng = np.random.default_rng(77)
    segs, radius = [], []
    for r in (3.0, 0.6):
        for scale in (1.0, 40.0, 95.0):
            pts = []
            for k, rel in enumerate((1e-5, 1e-6, 3e-7, 1e-7, 1e-8, 1e-10, 1e-13, 1e-15, 0.0)):
                for sign in (-1.0, 1.0):
                    c = rng.uniform(-1, 1, 3) * scale + np.array([400.0 * (2 * k + (sign > 0)), 0, 0]) * (scale / 95.0 + 0.2)
                    u = rng.normal(0, 1, 3)
                    u /= np.linalg.norm(u)
                    d = r * (1.0 + sign * rel)
                    pts += [c, c + u * d]                 # an isolated pair at (almost exactly) distance d
            pts = np.array(pts)
            segs.append(pts[rng.permutation(len(pts))])
            radius.append(r)
    radius = np.array(radius)
    for nb in (1,):
        P, base, cnt, cap = _pool_from_segments(segs)
        px, py, pz = _t(P[:, 0]), _t(P[:, 1]), _t(P[:, 2])
        S = len(segs)
        seg_cnt = _t(cnt)
        st.radius_filter(px, py, pz, _t(base), seg_cnt, _t(radius), nb, S, cap,
                         torch.zeros(S + 1, dtype=torch.int32, device=DEV),
                         torch.zeros(cap, dtype=torch.uint8, device=DEV),
                         torch.zeros(st.rf_queue_ints(cap), dtype=torch.int32, device=DEV))
        torch.cuda.synchronize()
        out_cnt = seg_cnt.cpu().numpy()
        X = torch.stack([px, py, pz], 1).cpu().numpy()
        n_kept = n_drop = 0
        for s, pts in enumerate(segs):
            keep = O.radius_outlier(pts, nb, radius[s])
            n_kept += len(keep)
            n_drop += len(pts) - len(keep)
            assert out_cnt[s] == len(keep), (s, out_cnt[s], len(keep))
            assert np.array_equal(X[base[s]:base[s] + out_cnt[s]], pts[keep]), s
        assert n_kept > 20 and n_drop > 20            # both outcomes occur right at the threshold


# ------------------------------------------------------------------ a12
def test_ballquery_fuse_matches_oracle(st):
    rng = np.random.default_rng(12)
    cases = [(0, 10), (10, 0), (40, 300), (1, 50), (1500, 2600), (3, 3), (4500, 900), (4096, 5000)]
    segsA, segsB = [], []
    for na, nb in cases:
        a = rng.normal(0, 2.0, (na, 3))
        b = rng.normal(0, 2.0, (nb, 3))
        if na and nb:
            h = nb // 2
            b[:h] = a[rng.integers(0, na, h)] + rng.normal(0, 0.06, (h, 3))
        segsA.append(a)
        segsB.append(b)
    # pool layout: A_s then B_s adjacent (as dfu3d_segments_build lays them out)
    chunks, base_a, base_b, cur = [], [], [], 0
    for a, b in zip(segsA, segsB):
        base_a.append(cur); cur += len(a)
        base_b.append(cur); cur += len(b)
        chunks += [a, b]
    P = np.concatenate(chunks)
    cap = len(P) + 16
    Pp = np.full((cap, 3), 9.0); Pp[:len(P)] = P
    px, py, pz = _t(Pp[:, 0]), _t(Pp[:, 1]), _t(Pp[:, 2])
    S = len(cases)
    cnt_a = _t(np.array([len(a) for a in segsA], np.int32))
    cnt_b = _t(np.array([len(b) for b in segsB], np.int32))
    tb = _t(np.array(base_b, np.int64))
    st.ballquery_fuse(px, py, pz, _t(np.array(base_a, np.int64)), cnt_a, tb, cnt_b, 0.1, S, cap,
                      torch.zeros(2 * S + 2, dtype=torch.int32, device=DEV),
                      torch.zeros(cap, dtype=torch.uint8, device=DEV))
    torch.cuda.synchronize()
    X = torch.stack([px, py, pz], 1).cpu().numpy()
    ncb = cnt_b.cpu().numpy()
    nbase = tb.cpu().numpy()
    for s, (a, b) in enumerate(zip(segsA, segsB)):
        keep = O.ball_query(b, a, 0.1) if len(a) and len(b) else np.ones(len(b), bool)
        assert ncb[s] == keep.sum(), s
        assert nbase[s] == base_a[s] + len(a)
        exp = np.concatenate([a, b[keep]])
        assert np.array_equal(X[base_a[s]:base_a[s] + len(exp)], exp), s


def test_masked_ballquery_equals_filter_then_fuse(st):
    """radius filter without its compaction + dfu3d_ballquery_fuse_masked == the two stages one after the
    other (my_loader.py:587-605), checked against the oracle's filter and fuse."""
    rng = np.random.default_rng(121)
    cases = [(0, 40), (30, 0), (60, 700), (900, 4000), (5, 5), (4200, 3000)]
    segsA, segsB = [], []
    for na, nb in cases:
        a = rng.normal(0, 2.0, (na, 3))
        b = rng.normal(0, 2.0, (nb, 3))
        if na and nb:
            h = nb // 2
            b[:h] = a[rng.integers(0, na, h)] + rng.normal(0, 0.06, (h, 3))
            far = rng.random(nb) < 0.2                              # isolated pseudo points: the filter drops them
            b[far] = rng.uniform(-300, 300, (int(far.sum()), 3))
        segsA.append(a)
        segsB.append(b)
    chunks, base_a, base_b, cur = [], [], [], 0
    for a, b in zip(segsA, segsB):
        base_a.append(cur); cur += len(a)
        base_b.append(cur); cur += len(b)
        chunks += [a, b]
    P = np.concatenate(chunks)
    cap = len(P) + 16
    Pp = np.full((cap, 3), 9.0); Pp[:len(P)] = P
    px, py, pz = _t(Pp[:, 0]), _t(Pp[:, 1]), _t(Pp[:, 2])
    S = len(cases)
    cnt_a = _t(np.array([len(a) for a in segsA], np.int32))
    cnt_b = _t(np.array([len(b) for b in segsB], np.int32))
    ta, tb = _t(np.array(base_a, np.int64)), _t(np.array(base_b, np.int64))
    radius = _t(np.array([0.6, 3.0, 0.6, 3.0, 0.6, 3.0]))
    tile_off = torch.zeros(2 * S + 2, dtype=torch.int32, device=DEV)
    flags = torch.zeros(cap, dtype=torch.uint8, device=DEV)
    queue = torch.zeros(st.rf_queue_ints(cap), dtype=torch.int32, device=DEV)
    st.radius_filter(px, py, pz, tb, cnt_b, radius, 1, S, cap, tile_off, flags, queue,
                     phases=st.RF_ALL & ~st.RF_COMPACT)
    st.ballquery_fuse(px, py, pz, ta, cnt_a, tb, cnt_b, 0.1, S, cap, tile_off, flags, masked=True)
    torch.cuda.synchronize()
    X = torch.stack([px, py, pz], 1).cpu().numpy()
    ncb = cnt_b.cpu().numpy()
    r = radius.cpu().numpy()
    for s, (a, b) in enumerate(zip(segsA, segsB)):
        b1 = b[O.radius_outlier(b, 1, r[s])] if len(b) else b
        keep = O.ball_query(b1, a, 0.1) if len(a) and len(b1) else np.ones(len(b1), bool)
        exp = np.concatenate([a, b1[keep]])
        assert ncb[s] == keep.sum(), s
        assert np.array_equal(X[base_a[s]:base_a[s] + len(exp)], exp), s


# ------------------------------------------------------------------ a11
def test_stat_filter_matches_oracle(st):
    rng = np.random.default_rng(11)
    segs = [_clustered_points(rng, n, 0.5, 0.1) for n in (0, 1, 2, 31, 400, 1300)]
    P, base, cnt, cap = _pool_from_segments(segs)
    px, py, pz = _t(P[:, 0]), _t(P[:, 1]), _t(P[:, 2])
    S = len(segs)
    seg_cnt = _t(cnt)
    enable = np.ones(S, np.int32); enable[3] = 0
    st.stat_filter(px, py, pz, _t(base), seg_cnt, _t(enable), 30, 0.3, S, cap,
                   torch.zeros(S + 1, dtype=torch.int32, device=DEV),
                   torch.zeros(cap, dtype=torch.uint8, device=DEV),
                   torch.zeros(cap, dtype=torch.float64, device=DEV))
    torch.cuda.synchronize()
    X = torch.stack([px, py, pz], 1).cpu().numpy()
    out = seg_cnt.cpu().numpy()
    for s, pts in enumerate(segs):
        keep = O.statistical_outlier(pts, 30, 0.3) if enable[s] else np.arange(len(pts))
        assert out[s] == len(keep), (s, out[s], len(keep))
        assert np.array_equal(X[base[s]:base[s] + out[s]], pts[keep]), s


# ------------------------------------------------------------------ a13
def _cluster_cases(rng):
    cases = []
    for n in (1, 2, 3, 17, 130, 700, 2500):
        k = max(1, n // 60)
        ctr = rng.uniform(-40, 40, (k, 2))
        cases.append(ctr[rng.integers(0, k, n)] + rng.normal(0, 0.8, (n, 2)))
    cases.append(np.stack([np.arange(40) * 2.99 + 1.0, np.zeros(40)], 1)[rng.permutation(40)])   # chain
    cases.append(np.stack([np.arange(40) * 3.2 + 1.0, np.zeros(40)], 1))                        # all apart
    cases.append(np.zeros((0, 2)))
    cases.append(rng.uniform(-60, 60, (13000, 2)))               # large-LDS variant
    two = np.concatenate([rng.normal(0, 1.0, (2500, 2)), rng.normal(0, 1.0, (2600, 2)) + [40.0, 0.0]])
    cases.append(two[rng.permutation(len(two))])                 # two dense blobs, interleaved
    cases.append(np.concatenate([rng.normal(0, 2.0, (29000, 2)), rng.uniform(-200, 200, (600, 2))]))  # 16-bit LDS parents
    cases.append(np.concatenate([rng.normal(0, 3.0, (61000, 2)), rng.uniform(-300, 300, (700, 2))]))  # parents in global memory
    # two dense bands whose closest pair is one bridge at a gap around R (3.0 + 0.001 |p| = 3.0104 there): every cell pair
    # across the gap passes the box test and must be decided point by point (strip pruning), on either side of R
    for gap in (3.0, 3.0100, 3.01040, 3.01044, 3.01048, 3.0106, 3.02, 3.3):
        a = np.stack([rng.uniform(0, 20, 3000), rng.uniform(-0.5, -0.02, 3000)], 1)
        b = np.stack([rng.uniform(0, 20, 3000), gap + rng.uniform(0.02, 0.5, 3000)], 1)
        bridge = np.array([[10.0, 0.0], [10.0, gap]])
        pts = np.concatenate([a, b, bridge])
        cases.append(pts[rng.permutation(len(pts))])
    return cases


def test_range_cluster_matches_oracle(st):
    rng = np.random.default_rng(13)
    cases = _cluster_cases(rng)
    segs = [np.concatenate([c, np.zeros((len(c), 1))], 1) for c in cases]
    P, base, cnt, cap = _pool_from_segments(segs)
    px, py = _t(P[:, 0]), _t(P[:, 1])
    S = len(segs)
    label = torch.full((cap,), -7, dtype=torch.int32, device=DEV)
    st.range_cluster(px, py, _t(base), _t(cnt), S, 3.0, 0.001, label, cap)
    torch.cuda.synchronize()
    lab = label.cpu().numpy()
    for s, c in enumerate(cases):
        exp = O.range_cluster_labels(c[:, 0], c[:, 1], 3.0, 0.001)
        assert np.array_equal(lab[base[s]:base[s] + len(c)], exp), s


# ------------------------------------------------------------------ a14/a15
def test_lshape_fit_matches_oracle(st):
    from dfu3d_amd import synth
    from dfu3d_amd.params import Params
    rng = np.random.default_rng(14)
    p = Params()
    n_theta, dtheta = p.thetas()
    assert n_theta == 89
    cal = _rand_calib(rng, 30.0)
    oc = _oracle_calib(cal)
    M = 4
    segs, classes, iscar, boxes = [], [], [], []
    def lshape(cx, cy, L, Wd, yaw, n):
        k = n // 2
        e1 = np.stack([rng.uniform(-L / 2, L / 2, k), np.full(k, -Wd / 2)], 1)
        e2 = np.stack([np.full(n - k, -L / 2), rng.uniform(-Wd / 2, Wd / 2, n - k)], 1)
        q = np.vstack([e1, e2]) + rng.normal(0, 0.02, (n, 2))
        R = np.array([[math.cos(yaw), -math.sin(yaw)], [math.sin(yaw), math.cos(yaw)]])
        return (q @ R.T + np.array([cx, cy]))[rng.permutation(n)]
    shapes = [lshape(12, 3, 4.6, 1.9, 0.4, 90), lshape(-20, 8, 7.0, 2.5, 1.2, 300),
              np.concatenate([lshape(8, -5, 4.5, 1.8, 0.1, 60), lshape(25, 14, 4.2, 1.9, 2.0, 45)]),
              rng.normal(0, 0.2, (2, 2)) + 6, rng.normal(0, 0.1, (1, 2)) + 9,
              np.stack([np.linspace(3, 23, 70), np.full(70, 2.0)], 1),    # 20 m x 0 -> Car aspect skip
              lshape(15, -9, 4.4, 1.8, -0.9, 4000), np.zeros((0, 2))]
    for i, xy in enumerate(shapes):
        z = rng.uniform(-1.7, 0.3, len(xy))
        segs.append(np.concatenate([xy, z[:, None]], 1))
        classes.append(i % 10)
        iscar.append(1 if i % 10 == 0 or i == 5 else 0)
        boxes.append([100.0 + i, 50.5, 300.25, 200.0])
    while len(segs) % M:
        segs.append(np.zeros((0, 3))); classes.append(0); iscar.append(0); boxes.append([0, 0, 0, 0])
    S = len(segs)
    V = S // M
    P, base, cnt, cap = _pool_from_segments(segs)
    px, py, pz = _t(P[:, 0]), _t(P[:, 1]), _t(P[:, 2])
    tb, tc = _t(base), _t(cnt)
    label = torch.zeros(cap, dtype=torch.int32, device=DEV)
    st.range_cluster(px, py, tb, tc, S, p.R0, p.Rd, label, cap)
    cap_rows = 64
    rows = torch.zeros(cap_rows * st.ROW_DOUBLES, dtype=torch.float64, device=DEV)
    n_rows = torch.zeros(1, dtype=torch.int32, device=DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    calib = _t(np.stack([cal.record()] * V))
    st.lshape_fit(px, py, pz, label, tb, tc, S, M, calib, _t(np.array(classes, np.int32)),
                  _t(np.array(iscar, np.int32)), _t(np.array(boxes, np.float32)),
                  torch.ones(S, dtype=torch.float32, device=DEV), n_theta, dtheta,
                  p.car_aspect_max, torch.zeros(cap, dtype=torch.float64, device=DEV),
                  torch.zeros(cap, dtype=torch.float64, device=DEV),
                  torch.zeros(cap, dtype=torch.int32, device=DEV), cap_rows, rows, n_rows,
                  status, cap)
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    n = int(n_rows.item())
    R = rows.view(cap_rows, st.ROW_DOUBLES)[:n].cpu().numpy()
    R = R[np.lexsort((R[:, 2], R[:, 1], R[:, 0]))]
    exp = []
    for s, pts in enumerate(segs):
        name = "Car" if iscar[s] else "Truck"
        for r in O.generate_anns(name, pts, classes[s], np.array(boxes[s], np.float32), oc, O.Params(), inst=s % M):
            exp.append((s // M, s % M, r.cluster, r))
    assert n == len(exp), (n, len(exp))
    for got, (v, j, k, r) in zip(R, exp):
        assert (int(got[0]), int(got[1]), int(got[2]), int(got[3])) == (v, j, k, r.cls)
        np.testing.assert_allclose(got[4:16], r.as_vector(), rtol=1e-9, atol=1e-9)


def test_lshape_fit_heading_ties_are_decided_like_the_reference(st):
    """The heading search is two-tier (two-sweep variances for all 89 headings, the reference's three sweeps only for
    the headings within 1e-8 of the best one): point sets with EXACT symmetries -- several headings share the best
    cost, or differ from it in the last bits -- must still come out with the reference's arg-max (first strict
    maximum, rectangle_fitting.py:135-136).  One case per kernel of the stage (<= 64 points: a lane per heading;
    <= 2048: a wave per heading; larger: heading batches), plus shapes whose cost does not depend on the heading."""
    from dfu3d_amd.params import Params
    rng = np.random.default_rng(41)
    p = Params()
    n_theta, dtheta = p.thetas()
    cal = _rand_calib(rng, 30.0)
    oc = _oracle_calib(cal)
    M = 4

    def square_grid(k, side, cx, cy):                      # k x k lattice: invariant under 90-degree rotations and mirrors
        g = (np.arange(k) - (k - 1) / 2.0) * (side / max(k - 1, 1))
        xx, yy = np.meshgrid(g, g)
        return np.stack([xx.ravel() + cx, yy.ravel() + cy], 1)

    def ring(n, r, cx, cy):                                # regular n-gon: cost is the same for many headings
        a = 2 * np.pi * np.arange(n) / n
        return np.stack([cx + r * np.cos(a), cy + r * np.sin(a)], 1)

    def mirrored_l(n, cx, cy):                             # an L shape and its mirror image about the diagonal: theta <-> 90 - theta
        k = n // 4
        t = rng.uniform(0.0, 4.0, k)
        a = np.stack([t, np.zeros(k)], 1)
        b = np.stack([np.zeros(k), rng.uniform(0.0, 1.5, k)], 1)
        q = np.vstack([a, b])
        return np.vstack([q, q[:, ::-1]]) + np.array([cx, cy])
    shapes = [square_grid(7, 2.0, 10.0, 4.0),              # 49 points  -> k_fit_tiny
              ring(60, 1.5, -12.0, 6.0),
              square_grid(30, 2.5, 14.0, -7.0),            # 900 points -> k_fit_medium
              ring(720, 2.0, 20.0, 11.0),
              mirrored_l(400, -18.0, -9.0),
              square_grid(60, 2.8, 25.0, 3.0),             # 3600 points -> k_fit_big_*
              mirrored_l(5000, 8.0, 16.0),
              np.array([[3.0, 3.0], [4.0, 4.0]])]          # two points: every heading but 45 degrees has zero variance
    segs, classes, iscar, boxes = [], [], [], []
    for i, xy in enumerate(shapes):
        z = rng.uniform(-1.7, 0.3, len(xy))
        segs.append(np.concatenate([xy, z[:, None]], 1))
        classes.append(i % 10); iscar.append(0); boxes.append([10.0 + i, 5.5, 30.25, 20.0])
    while len(segs) % M:
        segs.append(np.zeros((0, 3))); classes.append(0); iscar.append(0); boxes.append([0, 0, 0, 0])
    S = len(segs)
    V = S // M
    P, base, cnt, cap = _pool_from_segments(segs)
    px, py, pz = _t(P[:, 0]), _t(P[:, 1]), _t(P[:, 2])
    tb, tc = _t(base), _t(cnt)
    label = torch.zeros(cap, dtype=torch.int32, device=DEV)
    st.range_cluster(px, py, tb, tc, S, p.R0, p.Rd, label, cap)
    cap_rows = 64
    rows = torch.zeros(cap_rows * st.ROW_DOUBLES, dtype=torch.float64, device=DEV)
    n_rows = torch.zeros(1, dtype=torch.int32, device=DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    st.lshape_fit(px, py, pz, label, tb, tc, S, M, _t(np.stack([cal.record()] * V)), _t(np.array(classes, np.int32)),
                  _t(np.array(iscar, np.int32)), _t(np.array(boxes, np.float32)),
                  torch.ones(S, dtype=torch.float32, device=DEV), n_theta, dtheta,
                  p.car_aspect_max, torch.zeros(cap, dtype=torch.float64, device=DEV),
                  torch.zeros(cap, dtype=torch.float64, device=DEV),
                  torch.zeros(cap, dtype=torch.int32, device=DEV), cap_rows, rows, n_rows, status, cap)
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    n = int(n_rows.item())
    R = rows.view(cap_rows, st.ROW_DOUBLES)[:n].cpu().numpy()
    R = R[np.lexsort((R[:, 2], R[:, 1], R[:, 0]))]
    exp = []
    for s_, pts in enumerate(segs):
        for r in O.generate_anns("Truck", pts, classes[s_], np.array(boxes[s_], np.float32), oc, O.Params(), inst=s_ % M):
            exp.append((s_ // M, s_ % M, r.cluster, r))
    assert n == len(exp) and n >= 8, (n, len(exp))
    for got, (v, j, k, r) in zip(R, exp):
        assert (int(got[0]), int(got[1]), int(got[2])) == (v, j, k)
        # column 18 of the engine row is the chosen heading: the oracle's (= the reference's) arg-max, exactly
        np.testing.assert_allclose(got[4:16], r.as_vector(), rtol=1e-9, atol=1e-9)


# ------------------------------------------------------------------ a4/a5/a6
def _lidar_setup(seed, n_frames=2, cams=3, H=225, W=400, M=4):
    from dfu3d_amd import synth
    from dfu3d_amd.params import Params
    scenes = [synth.make_scene(seed + f, H=H, W=W, M=M, cams=cams, dense=True, k_min=10, k_max=16)
              for f in range(n_frames)]
    p = Params(bounds_hw=(H, W), fov_hw=(H, W))
    return scenes, p


def test_fov_plane_label_match_oracle(st):
    from dfu3d_amd import synth
    scenes, p = _lidar_setup(20)
    H, W = p.bounds_hw
    b = synth.to_view_batch(scenes, p, DEV)
    V = b.view_frame.numel()
    M = b.masks.shape[1]
    cap_n = max(s.points.shape[0] for s in scenes)
    i32 = lambda n: torch.zeros(n, dtype=torch.int32, device=DEV)
    f64 = lambda n: torch.zeros(n, dtype=torch.float64, device=DEV)
    fov_idx, n_fov = i32(V * cap_n), i32(V)
    st.fov_filter(b.points, b.pt_off, b.view_frame, b.calib, V, p.fov_hw, cap_n, fov_idx, n_fov,
                  b.host_pt_off, b.host_view_frame)
    plane = f64(V * 4)
    st.plane_ransac(b.points, b.pt_off, b.view_frame, fov_idx, n_fov, V, cap_n, p.plane_max_hs,
                    p.plane_range, p.ransac_trials, p.ransac_seed, b.view_key, i32(V * cap_n), plane)
    ag_pt, ib_pix, n_ag, K = i32(V * cap_n), i32(V * cap_n), i32(V), i32(V)
    bits, ix, iy, iz = i32(V * cap_n), f64(V * cap_n), f64(V * cap_n), f64(V * cap_n)
    # oracle planes go in so that the integer comparison below is apples to apples
    op = O.Params(bounds_hw=p.bounds_hw, fov_hw=p.fov_hw)
    planes_o = []
    v = 0
    for f, s in enumerate(scenes):
        for c, cal in enumerate(s.calibs):
            oc = _oracle_calib(cal)
            lid, flag = O.fov_filter(s.points.numpy(), oc, p.fov_hw)
            planes_o.append(O.plane_ransac(lid[:, :3], op, key=v))
            v += 1
    planes_o = np.array(planes_o)
    torch.cuda.synchronize()
    np.testing.assert_allclose(plane.cpu().numpy().reshape(V, 4), planes_o, rtol=1e-9, atol=1e-9)
    st.project_label(b.points, b.pt_off, b.view_frame, b.calib, _t(planes_o.reshape(-1)), fov_idx,
                     n_fov, b.masks, b.n_inst, V, M, H, W, cap_n, p.plane_offset, p.plane_range,
                     ag_pt, ib_pix, n_ag, K, bits, ix, iy, iz)
    torch.cuda.synchronize()
    fov_idx_h, n_fov_h = fov_idx.cpu().numpy().reshape(V, cap_n), n_fov.cpu().numpy()
    n_ag_h, K_h = n_ag.cpu().numpy(), K.cpu().numpy()
    bits_h = bits.cpu().numpy().reshape(V, cap_n)
    xyz_h = torch.stack([ix, iy, iz], 1).cpu().numpy().reshape(V, cap_n, 3)
    v = 0
    for f, s in enumerate(scenes):
        pts = s.points.numpy()
        for c, cal in enumerate(s.calibs):
            oc = _oracle_calib(cal)
            lid, flag = O.fov_filter(pts, oc, p.fov_hw)
            assert np.array_equal(fov_idx_h[v, :n_fov_h[v]], np.nonzero(flag)[0]), v
            pm = O.above_plane(lid[:, :3], planes_o[v], p.plane_offset, p.plane_range)
            lm = lid[pm]
            rows, Ko = O.label_inheritance(lm, oc, s.masks[c].numpy(), p.bounds_hw)
            assert n_ag_h[v] == pm.sum() and K_h[v] == Ko, v
            assert np.array_equal(xyz_h[v, :Ko], lm[:Ko, :3].astype(np.float64)), v
            for j in range(int(s.n_inst[c])):
                got = np.nonzero((bits_h[v, :Ko] >> j) & 1)[0]
                assert np.array_equal(got, rows[j]), (v, j)
            v += 1


# ------------------------------------------------------------------ a7/a8/a9
def _bp_run(st, depth, cal, masks, max_points=100, max_voxels=1000000, key_axis=1, cap_vox=1 << 16):
    V, H, W = depth.shape
    geom, E = st.make_geom(max_points_per_voxel=max_points, max_voxels=max_voxels)
    table = torch.empty(V * E * st.TABLE_ENTRY_BYTES, dtype=torch.uint8, device=DEV)
    st.bin_table_init(table, V * E)
    pw, bw = st.backproject_scratch_words(V, H, W, cap_vox, max_points, geom)
    i32 = lambda n: torch.zeros(n, dtype=torch.int32, device=DEV)
    f64 = lambda n: torch.zeros(n, dtype=torch.float64, device=DEV)
    n_vox, vox_pix, bits = i32(V), i32(V * cap_vox), i32(V * cap_vox)
    x, y, z = f64(V * cap_vox), f64(V * cap_vox), f64(V * cap_vox)
    status = i32(1)
    M = masks.shape[1]
    calib = _t(np.stack([c.record() for c in cal]))
    n_inst = _t(np.full(V, M, np.int32))
    pix_bin, blk = i32(pw), i32(bw)
    out = []
    for rep in range(2):            # second run proves the table was left clean
        st.backproject_bin(_t(depth), calib, _t(masks), n_inst, V, M, H, W, geom, E, key_axis,
                           table, pix_bin, blk, cap_vox, n_vox, vox_pix, bits, x, y, z, status)
        torch.cuda.synchronize()
        out.append((n_vox.cpu().numpy().copy(), vox_pix.cpu().numpy().reshape(V, cap_vox).copy(),
                    bits.cpu().numpy().reshape(V, cap_vox).copy(),
                    torch.stack([x, y, z], 1).cpu().numpy().reshape(V, cap_vox, 3).copy(),
                    int(status.item())))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    return out[0]


def _bp_oracle(depth_v, cal, masks_v, max_points, max_voxels, key_axis):
    oc = _oracle_calib(cal)
    p = O.Params(max_points_per_voxel=max_points, max_voxels=max_voxels)
    d = depth_v.copy()
    rows, cols, pl = O.backproject(d, oc, p.depth_min)
    zk = pl[:, 2] < p.z_max
    p0, r0, c0 = pl[zk], rows[zk], cols[zk]
    rep = O.voxel_sample(p0, p0[:, key_axis], p)
    pix = r0[rep] * depth_v.shape[1] + c0[rep]
    bits = np.zeros(len(rep), np.int64)
    for j in range(masks_v.shape[0]):
        bits |= (masks_v[j][r0[rep], c0[rep]] > 0).astype(np.int64) << j
    return pix, p0[rep], bits


def test_backproject_repair_path_with_colliding_keys(st, monkeypatch):
    """Test build of the library whose packed min-(key|pixel) word keeps only 14 key bits: the cheap representative
    is wrong for a large share of the voxels, so k_bp_vox queues them and the exact repair (k_bp_rebin, k_ovf_*,
    k_bp_fix) produces the answer -- it must still be the oracle's, bit for bit."""
    from dfu3d_amd import _lib, synth
    L = _lib.load_variant("keybits14")
    monkeypatch.setattr(_lib, "_LIB", L)
    s = synth.make_scene(33, H=180, W=320, M=4, cams=2, dense=True, k_min=10, k_max=14)
    depth = s.depth.numpy().copy()
    depth[0, 100:110, :] = 7.5
    masks = s.masks.numpy()
    for key_axis in (1, 2):
        n_vox, vox_pix, bits, xyz, status = _bp_run(st, depth, s.calibs, masks, 100, 1000000, key_axis)
        assert status == 0
        for v in range(depth.shape[0]):
            pix, pts, ob = _bp_oracle(depth[v], s.calibs[v], masks[v], 100, 1000000, key_axis)
            assert n_vox[v] == len(pix)
            assert np.array_equal(vox_pix[v, :len(pix)], pix), v
            assert np.array_equal(bits[v, :len(pix)], ob), v
            assert np.array_equal(xyz[v, :len(pix)], pts), v


def test_backproject_without_the_middle_tier_takes_the_full_fp64_paths(st, monkeypatch):
    """Test build of the library without the middle tier of the bin classification: every pixel the float32 tier
    leaves undecided is classified by k_bp_bin_amb (the reference's full fp64 expressions, scattered commits) -- in the
    product build that kernel sees a few pixels per launch, here it sees 0.7 % of them.  Same answer as the oracle's,
    bit for bit, and the same as the product build's."""
    from dfu3d_amd import _lib, synth
    s = synth.make_scene(35, H=180, W=320, M=4, cams=3, dense=True, k_min=10, k_max=14)
    depth = s.depth.numpy().copy()
    depth[0, 100:110, :] = 7.5
    masks = s.masks.numpy()
    ref = _bp_run(st, depth, s.calibs, masks, 100, 1000000, 2)
    L = _lib.load_variant("no_mid")
    monkeypatch.setattr(_lib, "_LIB", L)
    for key_axis in (1, 2):
        n_vox, vox_pix, bits, xyz, status = _bp_run(st, depth, s.calibs, masks, 100, 1000000, key_axis)
        assert status == 0
        for v in range(depth.shape[0]):
            pix, pts, ob = _bp_oracle(depth[v], s.calibs[v], masks[v], 100, 1000000, key_axis)
            assert n_vox[v] == len(pix)
            assert np.array_equal(vox_pix[v, :len(pix)], pix), v
            assert np.array_equal(bits[v, :len(pix)], ob), v
            assert np.array_equal(xyz[v, :len(pix)], pts), v
        if key_axis == 2:
            assert np.array_equal(n_vox, ref[0]) and np.array_equal(vox_pix, ref[1]) and np.array_equal(xyz, ref[3])


@pytest.mark.parametrize("max_points,max_voxels,key_axis", [(100, 1000000, 1), (100, 1000000, 2),
                                                            (3, 1000000, 1), (1, 1000000, 2),
                                                            (100, 500, 1), (2, 300, 1)])
def test_backproject_bin_matches_oracle(st, max_points, max_voxels, key_axis):
    from dfu3d_amd import synth
    s = synth.make_scene(31, H=180, W=320, M=4, cams=3, dense=True, k_min=10, k_max=14)
    depth = s.depth.numpy().copy()
    # degenerate rows: constant depth (exact key ties) and a flat near-camera patch
    depth[0, 100:110, :] = 7.5
    depth[1, 60:90, 100:200] = 0.02          # everything lands in a handful of bins
    masks = s.masks.numpy()
    n_vox, vox_pix, bits, xyz, status = _bp_run(st, depth, s.calibs, masks, max_points, max_voxels, key_axis)
    assert status == 0
    for v in range(depth.shape[0]):
        pix, pts, ob = _bp_oracle(depth[v], s.calibs[v], masks[v], max_points, max_voxels, key_axis)
        assert n_vox[v] == len(pix), (v, n_vox[v], len(pix))
        assert np.array_equal(vox_pix[v, :len(pix)], pix), v
        assert np.array_equal(bits[v, :len(pix)], ob), v
        assert np.array_equal(xyz[v, :len(pix)], pts), v       # back-projection bit-exact (fp64 FMA chain == numpy's dgemm)
