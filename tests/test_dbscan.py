"""RT-DBSCAN: the spec of oracle/dbscan_oracle.c against sklearn, and
the HIP path against both.  The reference has no source for this, so parity is unpinned; sklearn's
labelling is the external anchor."""
import os

import numpy as np
import pytest

import oracle
from owlraytracing_amd import datasets


def _boundary_free(xyz, eps, rel=1e-6):
    """True if no pair of points has a distance within rel*eps of eps (float64 check), i.e. fp32
    and float64 distance arithmetic must agree on every neighbourhood."""
    from scipy.spatial import cKDTree
    t = cKDTree(xyz.astype(np.float64))
    near = t.query_pairs(eps * (1 + rel), output_type="ndarray")
    d = np.linalg.norm(xyz[near[:, 0]].astype(np.float64) - xyz[near[:, 1]].astype(np.float64), axis=1)
    return not np.any(np.abs(d - eps) <= rel * eps)


def _pick_eps(xyz, eps):
    """nudge eps until no pair distance is within rounding of it (so sklearn's float64 agrees)"""
    for j in range(40):
        e = float(np.float32(eps * (1 + 1e-3 * j)))
        if _boundary_free(xyz, e):
            return e
    return None


def _cases():
    yield "blobs3d", datasets.gaussian_mixture3d(4000, components=6, sigma=0.02, seed=3), 0.012, 5
    yield "uniform_sparse", datasets.uniform3d(3000, seed=4), 0.03, 4
    yield "planar_taxi", datasets.pad_to_3d(datasets.taxi_like2d(4000, components=12, seed=5)), 0.004, 6
    yield "minpts1", datasets.uniform3d(800, seed=6), 0.05, 1
    yield "all_noise", datasets.uniform3d(500, seed=7), 0.001, 3


@pytest.mark.parametrize("name,xyz,eps,min_pts", list(_cases()), ids=[c[0] for c in _cases()])
def test_spec_equals_sklearn_labelling(name, xyz, eps, min_pts):
    from sklearn.cluster import DBSCAN
    eps32 = _pick_eps(xyz, eps)
    if eps32 is None:
        pytest.skip("no eps near the requested one is free of boundary pairs")
    ref = oracle.dbscan(xyz, eps32, min_pts)
    sk = DBSCAN(eps=eps32, min_samples=min_pts, algorithm="brute").fit(xyz.astype(np.float64))
    core = np.zeros(len(xyz), bool)
    core[sk.core_sample_indices_] = True
    assert np.array_equal(ref["core"], core)
    assert np.array_equal(ref["labels"], sk.labels_)  # identical numbering, not just a permutation
    assert ref["clusters"] == (sk.labels_.max() + 1 if (sk.labels_ >= 0).any() else 0)


def test_spec_invariants():
    xyz = datasets.gaussian_mixture3d(3000, components=5, sigma=0.03, seed=9)
    r = oracle.dbscan(xyz, 0.02, 4)
    lab, core = r["labels"], r["core"]
    assert np.all(lab[core] >= 0), "core points are never noise"
    # cluster ids follow the smallest core index of each cluster
    firsts = [np.flatnonzero(core & (lab == c))[0] for c in range(r["clusters"])]
    assert firsts == sorted(firsts)
    assert np.all(r["counts"] >= 1)
    assert np.array_equal(core, r["counts"] >= 4)


@pytest.mark.gpu
@pytest.mark.parametrize("name,xyz,eps,min_pts", list(_cases()), ids=[c[0] for c in _cases()])
def test_hip_dbscan_equals_the_spec(name, xyz, eps, min_pts):
    from owlraytracing_amd.trueknn import TrueKNN
    eps32 = float(np.float32(eps))
    ref = oracle.dbscan(xyz, eps32, min_pts)
    eng = TrueKNN()
    eng.build(xyz)
    got = eng.dbscan(eps32, min_pts, want_counts=True)
    assert np.array_equal(got["core"].cpu().numpy(), ref["core"])
    assert np.array_equal(got["counts"].cpu().numpy(), ref["counts"])
    assert np.array_equal(got["labels"].cpu().numpy(), ref["labels"])
    assert got["info"]["clusters"] == ref["clusters"]
    fast = eng.dbscan(eps32, min_pts)  # early exit of the core test must not change anything
    assert np.array_equal(fast["labels"].cpu().numpy(), ref["labels"])
    eng.close()


@pytest.mark.gpu
def test_hip_dbscan_larger_mixture_against_the_spec_and_sklearn():
    from sklearn.cluster import DBSCAN
    from owlraytracing_amd.trueknn import TrueKNN
    xyz = datasets.gaussian_mixture3d(200_000, components=64, sigma=0.02, seed=1)  # BASELINE config 3, scaled down
    eps, min_pts = float(np.float32(0.01)), 4
    ref = oracle.dbscan(xyz, eps, min_pts)
    eng = TrueKNN()
    eng.build(xyz)
    got = eng.dbscan(eps, min_pts)
    lab = got["labels"].cpu().numpy()
    assert np.array_equal(lab, ref["labels"])
    assert np.array_equal(got["core"].cpu().numpy(), ref["core"])
    sub = np.arange(0, len(xyz), 1)[:30000]
    if _boundary_free(xyz[sub], eps):
        sk = DBSCAN(eps=eps, min_samples=min_pts).fit(xyz[sub].astype(np.float64))
        r2 = oracle.dbscan(xyz[sub], eps, min_pts)
        assert np.array_equal(r2["labels"], sk.labels_)
    eng.close()


@pytest.mark.gpu
def test_group_unions_equal_point_unions(monkeypatch):
    """The union pass walks once per GROUP (maximal tight node); TKNN_DBSCAN_UNION=point selects the per-point walk it
    replaced.  Both must give the spec's labels -- also where groups face each other across a gap close to eps (probes that
    find nothing), on a lattice whose spacing is eps itself, and with exact duplicates."""
    from owlraytracing_amd.trueknn import TrueKNN
    rng = np.random.default_rng(11)
    slab = lambda m: rng.uniform(0, 1, (m, 3)).astype(np.float32) * np.float32([0.2, 0.2, 0.02])
    g = np.arange(24, dtype=np.float32) * np.float32(0.01)
    lattice = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3).astype(np.float32) + np.float32(0.37)
    dup = datasets.uniform3d(20000, seed=12)
    dup[rng.choice(20000, 8000, replace=False)] = dup[rng.integers(0, 20000, 8000)]
    cases = [
        ("gmm", datasets.gaussian_mixture3d(150_000, components=16, sigma=0.02, seed=2), 0.01, 4),
        ("slabs_apart", np.concatenate([slab(40000), slab(40000) + np.float32([0, 0, 0.0301])]), 0.0099, 4),
        ("slabs_joined", np.concatenate([slab(40000), slab(40000) + np.float32([0, 0, 0.0301])]), 0.0104, 4),
        ("lattice_at_eps", lattice, float(np.float32(0.01)), 3),
        ("lattice_below_eps", lattice, float(np.float32(0.01 * (1 - 1e-6))), 3),
        ("duplicates", dup, 0.02, 3),
    ]
    eng = TrueKNN()
    for name, xyz, eps, min_pts in cases:
        eps = float(np.float32(eps))
        ref = oracle.dbscan(xyz, eps, min_pts)
        eng.build(xyz)
        monkeypatch.delenv("TKNN_DBSCAN_UNION", raising=False)
        by_group = eng.dbscan(eps, min_pts)
        monkeypatch.setenv("TKNN_DBSCAN_UNION", "point")
        by_point = eng.dbscan(eps, min_pts)
        monkeypatch.delenv("TKNN_DBSCAN_UNION", raising=False)
        assert np.array_equal(by_group["labels"].cpu().numpy(), ref["labels"]), name
        assert np.array_equal(by_point["labels"].cpu().numpy(), ref["labels"]), name
        assert by_group["info"]["clusters"] == ref["clusters"] == by_point["info"]["clusters"], name
        assert by_group["info"]["node_tests"] < by_point["info"]["node_tests"], name
    eng.close()


@pytest.mark.gpu
def test_one_set_shortcut_and_side_stream_walks_change_nothing(monkeypatch):
    """Round 3: the second union pass drops tree nodes whose core points were ONE set when the first pass ended
    (db_uniform_kernel; TKNN_DB_UNIFORM=0: off), and the points that are not core walk for their core neighbours on a side
    stream beside the unions, leaving lists the label kernel reads (TKNN_DB_SIDE=0: the label kernel walks); a point's walk
    down to its group starts at the node over its wave's 64 slots (TKNN_DB_PATHS=0: at the root).  Labels, core
    flags and cluster counts must be the spec's either way -- on a mixture (clusters of one set, borders), on a set that is
    mostly noise (the lists do not fit their room: the label kernel walks by itself), with a minPts so large that a few
    listed points already overflow the room, and on slabs whose halves join only in the second pass."""
    from owlraytracing_amd.trueknn import TrueKNN
    rng = np.random.default_rng(21)
    slab = lambda m: rng.uniform(0, 1, (m, 3)).astype(np.float32) * np.float32([0.2, 0.2, 0.02])
    sparse = datasets.uniform3d(60_000, seed=22)
    sparse[:6000] = np.float32(0.5) + np.float32(0.01) * rng.standard_normal((6000, 3)).astype(np.float32)
    cases = [
        ("gmm", datasets.gaussian_mixture3d(200_000, components=16, sigma=0.02, seed=3), 0.01, 4),
        ("mostly_noise", sparse, 0.004, 4),
        ("large_min_pts", datasets.gaussian_mixture3d(60_000, components=8, sigma=0.03, seed=4), 0.02, 40),
        ("slabs_joined_late", np.concatenate([slab(40000), slab(40000) + np.float32([0, 0, 0.0271])]), 0.0104, 4),
        # blobs of 150 .. 900 points, each far smaller than eps: the node over a wave's 64 slots is tight, the group is one to
        # four levels above it (block paths: the walk to the group starts at that node's ancestors, or at the root)
        ("blobs_within_eps", np.concatenate([c + np.float32(0.0004) * rng.standard_normal((m, 3)).astype(np.float32)
                                             for c, m in zip(rng.uniform(0, 1, (60, 3)).astype(np.float32), rng.integers(150, 900, 60))]), 0.01, 4),
    ]
    eng = TrueKNN()
    for name, xyz, eps, min_pts in cases:
        eps = float(np.float32(eps))
        ref = oracle.dbscan(xyz, eps, min_pts)
        eng.build(xyz)
        for env in ({}, {"TKNN_DB_UNIFORM": "0"}, {"TKNN_DB_SIDE": "0"}, {"TKNN_DB_UNIFORM": "0", "TKNN_DB_SIDE": "0"}, {"TKNN_DB_PATHS": "0"}):
            for k in ("TKNN_DB_UNIFORM", "TKNN_DB_SIDE", "TKNN_DB_PATHS"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            for _ in range(2):  # (twice: the second call meets the side stream and the workspace of the first)
                got = eng.dbscan(eps, min_pts)
                assert got["info"]["clusters"] == ref["clusters"], (name, env)
                assert np.array_equal(got["labels"].cpu().numpy(), ref["labels"]), (name, env)
                assert np.array_equal(got["core"].cpu().numpy().astype(bool), ref["core"].astype(bool)), (name, env)
        for k in ("TKNN_DB_UNIFORM", "TKNN_DB_SIDE", "TKNN_DB_PATHS"):
            monkeypatch.delenv(k, raising=False)
    eng.close()


def _gap_quadruples(eps, split=0.25, rel=(-1e-5, 1e-5), steps=81):
    """Pairs of two-point groups facing each other along ONE axis across a gap of split * eps * (1 + d), d swept over
    `rel` -- the seam between the two passes of the group-union kernel (first pass: nearest faces within split * eps,
    second pass: the rest).  The facing points sit at 0 and at the gap on that axis, so fp32 resolves steps of 1e-7 of the
    gap.  Every quadruple is one cluster (the gap is a quarter of eps); quadruples are several eps apart."""
    eps = np.float32(eps)
    spacing = np.float32(3.0 + 6.0 * float(eps))
    width = np.float32(0.9) * eps  # a group's box: diagonal 0.9 eps < eps -> a tight node
    rows = []
    for j, d in enumerate(np.linspace(rel[0], rel[1], steps)):
        for axis in range(3):
            gap = np.float32(float(split) * float(eps) * (1.0 + d))
            base = spacing * np.float32([1 + j % 9, 1 + j // 9, 1 + j // 9])
            base[(axis + 1) % 3] = spacing * np.float32(1 + j % 9)
            base[(axis + 2) % 3] = spacing * np.float32(1 + j // 9)
            for t in (-width, np.float32(0), gap, gap + width):
                p = base.copy()
                p[axis] = t
                rows.append(p)
    return np.ascontiguousarray(np.stack(rows), dtype=np.float32)


def test_gap_quadruples_are_single_clusters_in_the_spec():
    for eps in (0.01, 1.0, 0.37, 0.0008):
        xyz = _gap_quadruples(eps)
        r = oracle.dbscan(xyz, float(np.float32(eps)), 2)
        assert r["clusters"] == len(xyz) // 4, eps
        assert np.array_equal(r["labels"], np.arange(len(xyz)) // 4), eps


@pytest.mark.gpu
def test_union_passes_leave_no_gap_between_them():
    """pass 1 of the group unions prefiltered with a per-axis reach of split * eps (1 + 1e-6) but accepted
    squared face distances up to split^2 eps^2 (1 + 1e-5), and pass 2 started above that bound: two groups whose faces were
    between the two numbers apart along one axis were united by neither pass.  The advisor's four points, and sweeps of such
    gaps along each axis at several eps (groups of two points, 0.9 eps wide, so each is a tight node)."""
    from owlraytracing_amd.trueknn import TrueKNN
    eng = TrueKNN()
    four = np.zeros((4, 3), np.float32)
    four[:, 0] = np.float32([0.0, 0.9, 1.1500007, 2.05])
    cases = [("advisor_four_points", four, 1.0, 2)]
    for eps in (0.01, 1.0, 0.37, 0.0008):
        cases.append(("gaps_eps_%g" % eps, _gap_quadruples(eps), eps, 2))
    for name, xyz, eps, min_pts in cases:
        eps = float(np.float32(eps))
        ref = oracle.dbscan(xyz, eps, min_pts)
        eng.build(xyz)
        got = eng.dbscan(eps, min_pts)
        assert got["info"]["clusters"] == ref["clusters"], name
        assert np.array_equal(got["labels"].cpu().numpy(), ref["labels"]), name
    eng.close()


@pytest.mark.gpu
def test_hip_dbscan_degenerate_sets():
    """Shapes the group machinery has to survive: one or two points, every point the same (one group of 50 000), an eps
    that makes the ROOT a tight node, NaN coordinates, minPts 1, and a tree without a single core point."""
    from owlraytracing_amd.trueknn import TrueKNN
    rng = np.random.default_rng(21)
    same = np.tile(np.float32([[0.3, 0.4, 0.5]]), (50_000, 1))
    with_nan = datasets.uniform3d(5000, seed=22)
    with_nan[rng.choice(5000, 40, replace=False), rng.integers(0, 3, 40)] = np.nan
    two_blobs = np.concatenate([same[:3000], same[:3000] + np.float32(0.25)])
    cases = [
        ("one_point", np.float32([[0.1, 0.2, 0.3]]), 0.01, 1),
        ("two_points_apart", np.float32([[0.1, 0.2, 0.3], [0.9, 0.9, 0.9]]), 0.01, 1),
        ("two_points_close", np.float32([[0.1, 0.2, 0.3], [0.1, 0.2, 0.3005]]), 0.01, 2),
        ("all_the_same", same, 0.01, 4),
        ("root_is_tight", datasets.uniform3d(3000, seed=23), 4.0, 5),
        ("nan_coordinates", with_nan, 0.05, 3),
        ("two_stacks_of_duplicates", two_blobs[rng.permutation(len(two_blobs))], 0.01, 10),
        ("no_core_point", datasets.uniform3d(2000, seed=24), 1e-4, 2),
    ]
    eng = TrueKNN()
    for name, xyz, eps, min_pts in cases:
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        eps = float(np.float32(eps))
        ref = oracle.dbscan(xyz, eps, min_pts)
        eng.build(xyz)
        got = eng.dbscan(eps, min_pts)
        assert np.array_equal(got["labels"].cpu().numpy(), ref["labels"]), name
        assert np.array_equal(got["core"].cpu().numpy(), ref["core"]), name
        assert got["info"]["clusters"] == ref["clusters"], name
    eng.close()


@pytest.mark.gpu
def test_hip_dbscan_argument_errors():
    from owlraytracing_amd import _lib
    from owlraytracing_amd.trueknn import TrueKNN
    eng = TrueKNN()
    with pytest.raises(_lib.TknnError):
        eng.n = 10
        eng.dbscan(0.1, 3)
    eng.build(datasets.uniform3d(100, seed=1))
    for eps, m in ((0.0, 3), (float("nan"), 3), (0.1, 0)):
        with pytest.raises(_lib.TknnError):
            eng.dbscan(eps, m)
    eng.close()


@pytest.mark.gpu
def test_hip_dbscan_rows_not_ids_and_the_assign_step():
    """An engine built with ids (sharded use) still indexes DBSCAN results by row, and
    tknnDbscanAssign reproduces the last step from caller-decided core labels."""
    import torch

    from owlraytracing_amd.trueknn import TrueKNN
    xyz = datasets.gaussian_mixture3d(30_000, components=5, sigma=0.03, seed=9)
    eps, min_pts = float(np.float32(0.015)), 6
    ref = oracle.dbscan(xyz, eps, min_pts)
    eng = TrueKNN()
    ids = (np.arange(len(xyz), dtype=np.int32)[::-1] * 7 + 1_000_000).copy()  # far outside 0..n-1
    eng.build(torch.from_numpy(xyz).cuda(), torch.from_numpy(ids).cuda())
    got = eng.dbscan(eps, min_pts)
    assert np.array_equal(got["core"].cpu().numpy(), ref["core"].astype(bool))
    assert np.array_equal(got["labels"].cpu().numpy(), ref["labels"])
    # relabel the clusters arbitrarily (order-reversing), hand in core labels only, get everything back
    ncl = int(ref["clusters"])
    remap = np.arange(ncl, dtype=np.int32)[::-1] * 3 + 5
    core_label = np.where(ref["core"].astype(bool), remap[np.maximum(ref["labels"], 0)], -1).astype(np.int32)
    out = eng.dbscan_assign(eps, torch.from_numpy(core_label).cuda()).cpu().numpy()
    core = ref["core"].astype(bool)
    assert np.array_equal(out[core], core_label[core])
    from scipy.spatial import cKDTree
    tree = cKDTree(xyz.astype(np.float64))
    for q in np.nonzero(~core)[0][:400]:
        want = -1
        for p in tree.query_ball_point(xyz[q].astype(np.float64), eps * 1.0001):
            d = xyz[p] - xyz[q]
            if core[p] and np.sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2], dtype=np.float32) <= np.float32(eps):
                want = core_label[p] if want < 0 else min(want, core_label[p])
        assert out[q] == want
    eng.close()


def test_threaded_spec_equals_the_serial_spec():
    """oracle.dbscan_threaded (bench.py's CPU baseline: same grid and arithmetic, lock-free unions on all cores)
    against the serial statement of the spec, label for label."""
    for name, xyz, eps, min_pts in _cases():
        eps32 = float(np.float32(eps))
        a, b = oracle.dbscan(xyz, eps32, min_pts), oracle.dbscan_threaded(xyz, eps32, min_pts)
        assert np.array_equal(a["labels"], b["labels"]) and np.array_equal(a["core"], b["core"]) and a["clusters"] == b["clusters"], name
    xyz = datasets.gaussian_mixture3d(120_000, components=64, sigma=0.02, seed=1)
    a, b = oracle.dbscan(xyz, float(np.float32(0.01)), 4), oracle.dbscan_threaded(xyz, float(np.float32(0.01)), 4)
    assert np.array_equal(a["labels"], b["labels"]) and np.array_equal(a["core"], b["core"])


@pytest.mark.gpu
def test_config3_full_size():
    """BASELINE config 3 at its full size -- 10 M Gaussian-mixture points, eps 0.01, minPts 4 (about 5 000
    neighbours per point: the tight-node path that the scaled-down sets never stress):
      * cluster count at 1 M points of the same mixture equals a single run of the serial spec;
      * at 10 M, core flags and labels of 2 500 sampled points recomputed from their eps-balls (KD-tree over
        all points, fp32 distance arithmetic of the spec): core <=> |N(p)| >= minPts; a core point shares its
        label with every core point in its ball; a border point carries the smallest label among the core points
        in its ball; a point with no core point in its ball is noise;
      * labels of all 10 M points equal the threaded CPU spec's (oracle.dbscan_threaded), as do the core flags."""
    import torch
    from scipy.spatial import cKDTree

    from owlraytracing_amd.trueknn import TrueKNN
    eps, min_pts = float(np.float32(0.01)), 4
    eng = TrueKNN()
    # 1 M: the whole spec, serially
    small = datasets.gaussian_mixture3d(1_000_000, components=64, sigma=0.02, seed=1)
    eng.build(small)
    got = eng.dbscan(eps, min_pts)
    ref = oracle.dbscan_threaded(small, eps, min_pts)
    assert got["info"]["clusters"] == ref["clusters"]
    assert np.array_equal(got["labels"].cpu().numpy(), ref["labels"])
    # 10 M
    n = 10_000_000
    xyz = datasets.gaussian_mixture3d(n, components=64, sigma=0.02, seed=1)
    eng.build(torch.from_numpy(xyz).cuda())
    got = eng.dbscan(eps, min_pts)
    info = got["info"]
    lab, core = got["labels"].cpu().numpy(), got["core"].cpu().numpy()
    assert info["clusters"] == int(lab.max()) + 1 and info["point_tests"] > 0 and info["union_ms"] > 0
    assert np.all(lab[core] >= 0)
    tree = cKDTree(xyz)  # float32 data: queries below re-test every returned pair with the spec's arithmetic
    rng = np.random.default_rng(33)
    sample = rng.choice(n, 2500, replace=False)
    balls = tree.query_ball_point(xyz[sample].astype(np.float64), eps * 1.0001, workers=-1)
    for q, ball in zip(sample, balls):
        ball = np.asarray(ball)
        d = xyz[ball] - xyz[q]
        dist = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2], dtype=np.float32)
        near = ball[dist <= np.float32(eps)]
        assert bool(core[q]) == (len(near) >= min_pts), q  # q itself is in `near`
        near_core = near[core[near]]
        if core[q]:
            assert np.all(lab[near_core] == lab[q]), q
        elif len(near_core):
            assert lab[q] == lab[near_core].min(), q
        else:
            assert lab[q] == -1, q
    full = oracle.dbscan_threaded(xyz, eps, min_pts)
    assert full["clusters"] == info["clusters"]
    assert np.array_equal(core, full["core"])
    assert np.array_equal(lab, full["labels"])
    eng.close()


def test_auto_grown_eps_spec():
    """dbref_dbscan_auto: the rounds are plain DBSCAN runs at eps0 * 2^t (fp32 doubling), the first whose noise is under
    the bound wins; noise never grows with eps; a generous eps0 takes one round."""
    xyz = datasets.pad_to_3d(datasets.taxi_like2d(8000, components=12, seed=5))
    eps0, min_pts = float(np.float32(0.0004)), 4
    noises, eps = [], np.float32(eps0)
    for _ in range(8):
        noises.append(int((oracle.dbscan(xyz, float(eps), min_pts)["labels"] < 0).sum()))
        eps = np.float32(eps * np.float32(2))
    assert all(a >= b for a, b in zip(noises, noises[1:])), noises
    for max_noise in (0.5, 0.1, 0.01):
        bound = int(np.floor(max_noise * len(xyz)))
        want = next(t for t, m in enumerate(noises) if m <= bound)
        r = oracle.dbscan_auto(xyz, eps0, min_pts, max_noise)
        assert r["rounds"] == want + 1 and r["noise"] == noises[want]
        assert r["eps"] == float(np.float32(eps0) * np.float32(2.0 ** want))
        plain = oracle.dbscan(xyz, r["eps"], min_pts)
        assert np.array_equal(r["labels"], plain["labels"]) and np.array_equal(r["core"], plain["core"])
    assert oracle.dbscan_auto(xyz, 0.05, min_pts, 0.01)["rounds"] == 1
    with pytest.raises(oracle.OracleError):
        oracle.dbscan_auto(xyz, 1e-7, min_pts, 0.0, max_rounds=3)


@pytest.mark.gpu
def test_hip_dbscan_noise_flags_equal_the_spec():
    """tknnDbscanNoise (one growth round of the auto-eps loop, for the sharded driver): the points tknnDbscan would label
    -1, by row -- also on an engine built with ids, and with NaN coordinates (noise by definition)."""
    from owlraytracing_amd.trueknn import TrueKNN
    xyz = datasets.pad_to_3d(datasets.taxi_like2d(30_000, components=10, seed=31))
    xyz[[5, 77, 4000], [0, 1, 2]] = np.nan
    eng = TrueKNN()
    for eps, min_pts, ids in [(0.0005, 4, None), (0.004, 6, np.arange(len(xyz), dtype=np.int32)[::-1].copy() + 1000)]:
        eps = float(np.float32(eps))
        ref = oracle.dbscan(xyz, eps, min_pts)
        eng.build(xyz, ids)
        got = eng.dbscan_noise(eps, min_pts)
        assert np.array_equal(got["noise"].cpu().numpy(), ref["labels"] < 0)
        assert got["count"] == int((ref["labels"] < 0).sum())
    with pytest.raises(Exception):
        eng.dbscan_noise(-1.0, 4)
    eng.close()


@pytest.mark.gpu
def test_hip_dbscan_auto_equals_the_spec():
    from owlraytracing_amd import _lib
    from owlraytracing_amd.trueknn import TrueKNN
    eng = TrueKNN()
    for name, xyz, eps0, min_pts, max_noise in (
            ("taxi2d", datasets.pad_to_3d(datasets.taxi_like2d(300_000, components=64, seed=2)), 0.00005, 4, 0.05),
            ("blobs3d", datasets.gaussian_mixture3d(200_000, components=32, sigma=0.02, seed=8), 0.0008, 6, 0.01),
            ("one_round", datasets.uniform3d(50_000, seed=4), 0.2, 3, 0.0)):
        eps0 = float(np.float32(eps0))
        ref = oracle.dbscan_auto(xyz, eps0, min_pts, max_noise)
        eng.build(xyz)
        got = eng.dbscan_auto(eps0, min_pts, max_noise)
        info = got["info"]
        assert (info["rounds"], info["eps"], info["noise"], info["clusters"]) == (ref["rounds"], ref["eps"], ref["noise"], ref["clusters"]), name
        assert np.array_equal(got["labels"].cpu().numpy(), ref["labels"]), name
        assert np.array_equal(got["core"].cpu().numpy(), ref["core"]), name
    with pytest.raises(_lib.TknnError):  # rounds run out
        eng.dbscan_auto(1e-9, 3, 0.0, max_rounds=2)
    for bad in ((0.0, 3, 0.1), (0.1, 0, 0.1), (0.1, 3, 1.5), (float("nan"), 3, 0.1)):
        with pytest.raises(_lib.TknnError):
            eng.dbscan_auto(*bad)
    eng.close()


def test_count_only_rounds_and_ball_check_equal_the_full_spec():
    """The cheap checkers used at sizes the full spec cannot be run at (BASELINE config 5's 50 M points; bench.py's spot
    check of config 3): the count-only growth loop reports the full loop's (rounds, eps, noise) and core flags; the
    sampled eps-ball check accepts the spec's own labelling and notices a wrong label, a wrong core flag, a border point
    given to another cluster and a noise point given a cluster."""
    xyz = datasets.pad_to_3d(datasets.taxi_like2d(8000, components=12, seed=5))
    eps0, min_pts = float(np.float32(0.0004)), 4
    for max_noise in (0.5, 0.1, 0.01):
        full, counts = oracle.dbscan_auto(xyz, eps0, min_pts, max_noise), oracle.dbscan_auto_counts(xyz, eps0, min_pts, max_noise)
        assert (full["rounds"], full["eps"], full["noise"]) == (counts["rounds"], counts["eps"], counts["noise"])
    for pts, eps, m in ((xyz, 0.004, 6), (datasets.gaussian_mixture3d(6000, components=5, sigma=0.03, seed=9), 0.02, 4)):
        eps = float(np.float32(eps))
        ref = oracle.dbscan(pts, eps, m)
        noise, core = oracle.dbscan_noise_count(pts, eps, m, want_core=True)
        assert noise == int((ref["labels"] < 0).sum()) and np.array_equal(core, ref["core"].astype(bool))
        every = np.arange(len(pts), dtype=np.int32)
        assert oracle.dbscan_ball_check(pts, eps, m, ref["labels"], ref["core"], every) == {"violations": 0, "first_bad": -1}
        lab, cr = ref["labels"].copy(), ref["core"].astype(bool).copy()
        border = np.flatnonzero(~cr & (lab >= 0))
        noise_pts = np.flatnonzero(lab < 0)
        core_pts = np.flatnonzero(cr)
        # one defect at a time, checked on the defective point itself
        for q, change in ((core_pts[3], "label"), (core_pts[7], "core"), (border[0] if len(border) else None, "label"),
                          (noise_pts[0] if len(noise_pts) else None, "label")):
            if q is None:
                continue
            l2, c2 = lab.copy(), cr.copy()
            if change == "label":
                l2[q] = l2[q] + 1 if l2[q] >= 0 else 0
            else:
                c2[q] = not c2[q]
            r = oracle.dbscan_ball_check(pts, eps, m, l2, c2, np.int32([q]))
            assert r["violations"] == 1 and r["first_bad"] == q, (q, change)


@pytest.mark.gpu
def test_config5_set_full_size_auto_eps():
    """BASELINE config 5's point set on ONE GPU -- 50 M heavy-tailed 2-D points (5 % exact duplicates), minPts 4, eps
    auto-grown from 5e-5 until at most 5 % of the points are noise (the run README / DESIGN quote a time for):
      * rounds, final eps and noise count equal the count-only CPU spec's growth loop (oracle.dbscan_auto_counts,
        all host cores), and the core flags of all 5e7 points equal that spec's at the final eps;
      * labels: noise count = number of -1 labels, clusters = max label + 1, core points are never noise, and the
        eps-balls of 2 500 sampled points recomputed by the CPU spec (oracle.dbscan_ball_check): core <=> |N(p)| >= minPts,
        a core point shares its label with every core point in its ball, a border point carries the smallest label among
        them, a point with none is noise."""
    import torch

    from owlraytracing_amd.trueknn import TrueKNN
    n = 50_000_000
    xyz = datasets.pad_to_3d(datasets.taxi_like2d(n, components=256, seed=2))
    eps0, min_pts, max_noise = float(np.float32(0.00005)), 4, 0.05
    eng = TrueKNN()
    eng.build(torch.from_numpy(xyz).cuda())
    got = eng.dbscan_auto(eps0, min_pts, max_noise)
    info = got["info"]
    lab, core = got["labels"].cpu().numpy(), got["core"].cpu().numpy().astype(bool)
    eng.close()
    torch.cuda.empty_cache()
    assert info["noise"] == int((lab < 0).sum()) <= int(np.floor(max_noise * n))
    assert info["clusters"] == int(lab.max()) + 1 and np.all(lab[core] >= 0)
    want = oracle.dbscan_auto_counts(xyz, eps0, min_pts, max_noise)
    assert (info["rounds"], info["eps"], info["noise"]) == (want["rounds"], want["eps"], want["noise"])
    noise, cpu_core = oracle.dbscan_noise_count(xyz, info["eps"], min_pts, want_core=True)
    assert noise == info["noise"] and np.array_equal(core, cpu_core)
    sample = np.random.default_rng(55).choice(n, 2500, replace=False).astype(np.int32)
    assert oracle.dbscan_ball_check(xyz, info["eps"], min_pts, lab, core, sample) == {"violations": 0, "first_bad": -1}
    print("config-5 set on one GPU: %d rounds, eps %.6g, noise %d, %d clusters; growth rounds %.1f ms + clustering %.1f ms" % (
        info["rounds"], info["eps"], info["noise"], info["clusters"], info["probe_ms"], info["solve_ms"]))


@pytest.mark.gpu
def test_rarely_taken_union_paths():
    """The group-union kernel's depth-first popping (a packet walk whose LDS stack is nearly full pops one reference at a
    time) and the host's fallback to per-point unions after a stack overflow never happen on ordinary inputs with the
    shipped stack of 512 references.  libowl_mi355x_diag.so is built with a stack of 320 (depth-first on these sets) and
    reports an overflow on demand (TKNN_DB_DIAG=16): scripts/db_fallback_check.py compares three sets with the CPU spec."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "owlraytracing_amd", "libowl_mi355x_diag.so")
    if not os.path.exists(lib):
        pytest.fail("libowl_mi355x_diag.so is not built: __graft_entry__.build() makes it (make -C owlraytracing_amd/csrc DIAG=1)")
    for diag, expect in ((None, "groups"), ("16", "groups 0 ")):
        env = dict(os.environ, OWL_MI355X_LIB=lib)
        env.pop("TKNN_DB_DIAG", None)
        if diag:
            env["TKNN_DB_DIAG"] = diag
        r = subprocess.run([sys.executable, os.path.join(root, "scripts", "db_fallback_check.py")], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("ok ")]
        assert len(lines) == 3, r.stdout
        if diag:  # the fallback ran: per-point unions report no groups and one union launch
            assert all(expect in ln and ln.rstrip().endswith("union launches 1") for ln in lines), r.stdout
        else:
            assert all("union launches 2" in ln for ln in lines), r.stdout


def _nested_clumps(rng, clumps, min_pts, eps):
    """Isolated clumps of fewer than minPts + 3 points whose offsets from the clump's centre shrink geometrically (the radix
    tree over a clump is a chain: a group five and more levels below the clump's node) and point in every direction, with
    half extents between 0.30 and 0.56 eps -- from the middle, where one point sits, the WHOLE clump is within eps while the
    clump's box has a diagonal above eps (no tight node).  scripts/db_twice_probe.py: the round-3 library gets 116 of
    98 335 core flags wrong on this generator at minPts = 33 (none with one-sided clumps, none at minPts <= 16)."""
    out = []
    for c in rng.uniform(0.05, 0.95, (clumps, 3)):
        m = int(rng.integers(max(3, min_pts // 2), min_pts + 3))
        h = eps * rng.uniform(0.30, 0.56)
        j = rng.integers(0, 9, (m, 1))
        pts = c + rng.choice([-1.0, 1.0], (m, 3)) * h * (0.5 ** j) * rng.uniform(0.7, 1.0, (m, 3))
        pts[0] = c
        out.append(pts)
    return np.concatenate(out).astype(np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("min_pts", [8, 16, 33, 64])
def test_core_count_never_counts_a_subtree_twice(min_pts, monkeypatch):
    """a point whose group cannot decide counts in the subtree four levels above its group first and then
    over the rest of the tree, stepping over that subtree -- but an ANCESTOR of the subtree that lies inside the eps-sphere
    used to be added whole, subtree included (6 + 7 = 13 >= minPts 10 for a point with 8 neighbours).  It takes minPts > 6,
    small unbalanced clumps of an extent between eps and 2 eps, and a point near the middle: none of the other suite cases.
    Core flags, counts and labels against the spec, with the walk to the group from the block paths and from the root.
    (minPts = 33 uses the very set on which the round-3 library fails.)"""
    from owlraytracing_amd.trueknn import TrueKNN
    rng = np.random.default_rng(7 * min_pts + len("two_sided"))
    eps = float(np.float32(0.004))
    xyz = _nested_clumps(rng, 20000 if min_pts <= 33 else 8000, min_pts, eps)
    ref = oracle.dbscan(xyz, eps, min_pts)
    assert 0 < ref["core"].sum() < len(xyz)  # both kinds of clumps are there
    eng = TrueKNN()
    eng.build(xyz)
    for paths in ("1", "0"):
        monkeypatch.setenv("TKNN_DB_PATHS", paths)
        got = eng.dbscan(eps, min_pts)
        wrong = np.flatnonzero(got["core"].cpu().numpy().astype(bool) != ref["core"].astype(bool))
        assert len(wrong) == 0, (paths, len(wrong), wrong[:5], ref["counts"][wrong[:5]])
        assert np.array_equal(got["labels"].cpu().numpy(), ref["labels"]), paths
        counted = eng.dbscan(eps, min_pts, want_counts=True)
        assert np.array_equal(counted["counts"].cpu().numpy(), ref["counts"]), paths
    monkeypatch.delenv("TKNN_DB_PATHS", raising=False)
    eng.close()
