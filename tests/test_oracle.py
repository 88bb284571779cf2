"""CPU tests of the checker itself: C restatement vs golden vectors vs numpy restatement."""
import numpy as np
import pytest

import oracle
from oracle.trueknn_numpy import distance32, trueknn_numpy
from owlraytracing_amd import datasets

from conftest import assert_rows_match


def test_neigh_layout_matches_reference_struct():
    # samples/s01-trueknn/GeomTypes.h:22-28: int, float, int, (pad), long long
    assert oracle.NEIGH_DTYPE.itemsize == 24
    assert oracle.NEIGH_DTYPE.fields["intersections"][1] == 16


def test_oracle_reproduces_golden(golden):
    r = oracle.trueknn(golden["xyz"], int(golden["k"]), float(golden["start_radius"]))
    assert r["rounds"] == int(golden["rounds"])
    assert np.float32(r["final_radius"]) == golden["final_radius"]
    assert_rows_match(r["idx"], r["dist"], r["intersections"], golden)
    assert np.all(r["num_neighbors"] == 0)
    # only slot 0 of a row carries state (deviceCode.cu:74,118): the other slots keep their init
    rows = r["fb"].reshape(len(golden["xyz"]), -1)
    if rows.shape[1] > 1:
        assert np.all(rows["numNeighbors"][:, 1:] == int(golden["k"]))
        assert np.all(rows["intersections"][:, 1:] == 0)


def test_numpy_restatement_agrees_with_golden(golden):
    if len(golden["xyz"]) > 2100:
        pytest.skip("covered when the fixtures are generated; keep the CPU suite short")
    r = trueknn_numpy(golden["xyz"], int(golden["k"]), float(golden["start_radius"]))
    assert r["rounds"] == int(golden["rounds"])
    assert np.array_equal(r["idx"], golden["idx"])
    assert np.array_equal(r["intersections"], golden["intersections"])
    assert np.array_equal(r["dist"], golden["dist"])


def test_row_invariants(golden):
    idx, dist, k = golden["idx"], golden["dist"], int(golden["k"])
    n = len(idx)
    assert np.all(np.diff(dist, axis=1) >= 0), "rows ascend"
    assert np.all(idx != np.arange(n)[:, None]), "no self"
    assert np.all((idx >= 0) & (idx < n))
    s = np.sort(idx, axis=1)
    assert np.all(s[:, 1:] != s[:, :-1]) if k > 1 else True, "no duplicate neighbours"
    # ties inside a row are ordered by index (canonical ascending visit order) -- unless the two
    # candidates entered the persistent list in different rounds, which the crossroundties fixture
    # is made of (deviceCode.cu:77-85,116-134)
    same = dist[:, 1:] == dist[:, :-1]
    if not golden["name"].startswith("crossroundties"):
        assert np.all(idx[:, 1:][same] > idx[:, :-1][same])
    else:
        assert np.any(idx[:, 1:][same] < idx[:, :-1][same])
    # recomputed distances are the stored ones, bit for bit
    q = np.repeat(np.arange(n), k)
    d = np.array([oracle.distance(golden["xyz"][p], golden["xyz"][qq])
                  for p, qq in zip(idx.ravel()[:500], q[:500])], np.float32)
    assert np.array_equal(d, dist.ravel()[:500])
    # every neighbour lies in the final box of its query: |c_p - q|_inf <= r_final (+rounding)
    rf = float(golden["final_radius"])
    linf = np.abs(golden["xyz"][idx] - golden["xyz"][:, None, :]).max(-1)
    assert np.all(linf <= rf * (1 + 1e-6) + 1e-30)


def test_visit_order_changes_only_ties():
    xyz = datasets.uniform3d(3000, seed=21)
    xyz[::7] = xyz[1::7][: len(xyz[::7])]  # force duplicates -> ties
    a = oracle.trueknn(xyz, 6, 0.01, order=oracle.ORDER_ASCENDING)
    for order, seed in ((oracle.ORDER_DESCENDING, 0), (oracle.ORDER_SHUFFLED, 5)):
        b = oracle.trueknn(xyz, 6, 0.01, order=order, seed=seed)
        assert b["rounds"] == a["rounds"]
        assert np.array_equal(a["dist"], b["dist"])
        assert np.array_equal(a["intersections"], b["intersections"])
        differs = np.any(a["idx"] != b["idx"], axis=1)
        # a row may differ only if it holds a tie, or ties with an excluded candidate at d_k
        for q in np.flatnonzero(differs)[:50]:
            assert set(a["idx"][q]) != set(b["idx"][q]) or len(np.unique(a["dist"][q])) < 6


def test_replay_and_numpy_restatement_agree_on_tie_heavy_sets():
    """Lattices with holes and coarsely quantised coordinates: most rows hold bit-identical distances,
    many between candidates first seen in different rounds.  The call-by-call replay (persistent
    lists, deviceCode.cu:77-85,116-134) and the independent numpy restatement (sort by distance,
    first round, index) must give the same rows index for index -- the rule the engines' tie pass
    is then held to (tests/test_trueknn_gpu.py::test_tie_heavy_sets_equal_the_replay)."""
    from oracle.trueknn_numpy import trueknn_numpy
    g = np.arange(9, dtype=np.float32) / np.float32(32)
    lattice = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    rng = np.random.default_rng(3)
    lattice = lattice[rng.random(len(lattice)) > 0.2]
    lattice = np.ascontiguousarray(lattice[rng.permutation(len(lattice))])
    quant = (np.round(datasets.uniform3d(1500, seed=8) * 32) / 32).astype(np.float32)
    crossed = 0
    for xyz, k, r0 in ((lattice, 5, 0.02), (lattice, 16, 0.02), (quant, 3, 0.02), (quant, 10, 0.015), (datasets.cross_round_ties(), 2, 1.0)):
        a = oracle.trueknn(xyz, k, r0)
        b = trueknn_numpy(xyz, k, r0)
        assert a["rounds"] == b["rounds"] >= 2
        assert np.array_equal(a["dist"].view(np.int32), b["dist"].view(np.int32))
        assert np.array_equal(a["intersections"], b["intersections"])
        assert np.array_equal(a["idx"], b["idx"])
        # rows where the plain (dist, index) order is NOT the answer: the round word matters there
        tie = a["dist"][:, 1:] == a["dist"][:, :-1]
        crossed += int(np.any(tie & (a["idx"][:, 1:] < a["idx"][:, :-1]), axis=1).sum())
    assert crossed > 0


def test_query_subset_equals_full_rows():
    xyz = datasets.uniform3d(5000, seed=4)
    full = oracle.trueknn(xyz, 5, datasets.start_radius(5000, 5))
    q = np.arange(0, 5000, 37, dtype=np.int32)
    sub = oracle.trueknn(xyz, 5, datasets.start_radius(5000, 5), query_ids=q)
    assert np.array_equal(sub["idx"][q], full["idx"][q])
    assert np.array_equal(sub["dist"][q], full["dist"][q])
    assert np.array_equal(sub["intersections"][q], full["intersections"][q])
    untouched = np.setdiff1d(np.arange(5000), q)
    assert np.all(sub["idx"][untouched] == -1)


def test_reference_would_not_terminate_for_n_le_k():
    with pytest.raises(oracle.OracleError):
        oracle.trueknn(datasets.uniform3d(5, seed=1), 5, 0.1, max_rounds=40)


def test_result_is_not_exact_knn_on_a_known_share_of_rows():
    # box-candidate kNN != exact kNN; the oracle must restate the former.
    xyz = datasets.uniform3d(20000, seed=0)
    k = 5
    r = oracle.trueknn(xyz, k, datasets.start_radius(20000, k))
    q = np.arange(0, 20000, 10, dtype=np.int32)
    bi, bd = oracle.bruteforce_knn(xyz, k, q)
    differ = np.any(np.sort(bi, 1) != np.sort(r["idx"][q], 1), axis=1).mean()
    assert 0.05 < differ < 0.35
    # but never better than exact: d_k(oracle) >= d_k(exact)
    assert np.all(r["dist"][q][:, -1] >= bd[:, -1])


def test_bruteforce_against_numpy():
    xyz = datasets.uniform3d(1500, seed=2)
    bi, bd = oracle.bruteforce_knn(xyz, 4)
    d = distance32(xyz[None, :, :].repeat(50, 0), xyz[:50, None, :])
    d[np.arange(50), np.arange(50)] = np.inf
    want = np.argsort(d, axis=1, kind="stable")[:, :4]
    assert np.array_equal(bi[:50], want)


def test_csv_reader_follows_reference_loop(tmp_path):
    pts = datasets.uniform3d(10, seed=3)
    p = tmp_path / "pts.csv"
    datasets.write_csv_points(str(p), pts)
    back = datasets.read_csv_points(str(p), 10, 3)
    assert np.array_equal(back, pts)
    assert len(datasets.read_csv_points(str(p), 4, 3)) == 4
    # 2-D file read as 2-D then padded with z = 0 (hostCode.cpp:115-118)
    text = "0.5,0.25\n1.5, 2.5\n3,4\n"
    two = datasets.read_csv_points(text, 3, 2)
    assert two.shape == (3, 2)
    assert np.array_equal(datasets.pad_to_3d(two)[:, 2], np.zeros(3, np.float32))
    # a non-numeric token ends its line (operator>> fails), like the reference's inner while
    assert len(datasets.read_csv_points("1,2,3\nx,5,6\n7,8,9\n", 3, 3)) == 2


def test_counter_based_points_do_not_depend_on_sharding():
    whole = datasets.uniform3d_counter(0, 3 * datasets.CHUNK // 2)
    a = datasets.uniform3d_counter(0, datasets.CHUNK - 5)
    b = datasets.uniform3d_counter(datasets.CHUNK - 5, 3 * datasets.CHUNK // 2)
    assert np.array_equal(np.concatenate([a, b]), whole)


def test_start_radius_sampler_is_seeded_and_sane():
    from owlraytracing_amd.radius import sample_start_radius
    xyz = datasets.uniform3d(5000, seed=12)
    r = sample_start_radius(xyz, n_samples=100, seed=3)
    assert r == sample_start_radius(xyz, n_samples=100, seed=3)
    assert 0 < r < 0.5
    # it is the smallest pairwise distance inside the sample
    pick = np.random.default_rng(3).choice(5000, 100, replace=False)
    s = xyz[pick].astype(np.float64)
    d = np.sqrt(((s[:, None] - s[None]) ** 2).sum(-1))
    d[np.arange(100), np.arange(100)] = np.inf
    assert abs(r - d.min()) < 1e-12
    # a solve started there terminates in a handful of rounds on uniform data
    assert oracle.trueknn(xyz, 5, r)["rounds"] <= 8


def test_tie_flag_bound_never_misses_a_cross_round_tie():
    """The team kernels drop a tie from the tie pass when `tie_may_straddle` (trueknn_team.hip) says
    that two candidates at that fp32 distance cannot have become candidates in different rounds.
    Restated here in float32 and checked against the literal box test of every round
    (deviceCode.cu:38-56): whenever two points at bit-identical distances from a query have
    different first rounds, the bound must have said "may straddle" -- in 3-D, in a plane (span
    sqrt(2)) and on a line, from far below the start radius to beyond the last one."""
    f = np.float32

    def may_straddle(d, r0, r_last, qmax, span):
        r = f(r0)
        while True:
            mg = f(f(qmax + f(f(2) * r)) * f(4.76837158203125e-07))
            if d <= f(f(r - mg) * f(0.99999)):
                return False
            if d <= f(f(r + mg) * f(span)):
                return True
            if not (r < r_last):
                return True
            r = f(r * f(2))

    def first_round(c, q, r0, rounds):
        r = f(r0)
        for lvl in range(rounds):
            lo, hi = (c - r).astype(f), (c + r).astype(f)
            if np.all((np.minimum(lo, hi) <= q) & (q <= np.maximum(lo, hi))):
                return lvl
            r = f(r * f(2))
        return rounds

    # integer triples with equal Euclidean and different Chebyshev norms; scaled by powers of two the
    # offsets, the differences and the squared distances are exact in fp32
    groups = [[(3, 4, 0), (5, 0, 0), (0, 3, 4)], [(1, 2, 2), (3, 0, 0), (2, 2, 1)], [(2, 3, 6), (7, 0, 0), (6, 2, 3)],
              [(1, 4, 8), (9, 0, 0), (4, 4, 7)], [(2, 6, 9), (11, 0, 0), (6, 6, 7)], [(10, 10, 5), (15, 0, 0), (2, 10, 11)]]
    planar = [[(3, 4, 0), (5, 0, 0), (4, 3, 0)], [(5, 12, 0), (13, 0, 0)], [(7, 24, 0), (25, 0, 0), (15, 20, 0)]]
    rng = np.random.default_rng(11)
    straddling = checked = 0
    for span, sets, dims in ((1.73206, groups, 3), (1.41422, planar, 2)):
        for _ in range(4000):
            grp = sets[rng.integers(len(sets))]
            scale = f(2.0 ** rng.integers(-12, 3))
            q = (rng.integers(-64, 64, 3) * scale * f(8)).astype(f)
            if dims == 2:
                q[2] = f(0.25)
            d_true = f(np.sqrt(f(sum(v * v for v in grp[0])))) * scale
            r0 = f(d_true * f(2.0 ** rng.uniform(-3.5, 1.2)))
            pts = []
            for t in grp:
                sign = rng.choice([-1, 1], 3)
                perm = rng.permutation(3) if dims == 3 else np.array([*rng.permutation(2), 2])
                off = (np.array(t, f)[perm] * sign * scale).astype(f)
                pts.append((q + off).astype(f))
            dist = [np.sqrt(((p - q).astype(f) ** 2).sum(dtype=f), dtype=f) for p in pts]
            assert all(x.view(np.int32) == dist[0].view(np.int32) for x in dist)
            firsts = [first_round(p, q, r0, 12) for p in pts]
            last = max(firsts)  # the query finishes no earlier than the round that sees all of them
            r_last = f(r0 * f(2.0 ** last))
            qmax = f(np.abs(q).max())
            checked += 1
            if len(set(firsts)) > 1:
                straddling += 1
                assert may_straddle(dist[0], r0, r_last, qmax, span), (q, pts, r0, firsts)
    assert straddling > 500 and checked == 8000


def test_per_query_radius_statement_is_the_reference_loop_per_query():
    """oracle.trueknn_per_query: with one radius for all it IS the reference solve; with classes of radii every class
    equals a plain solve of that class's queries (rows never depend on other rows)."""
    pts = datasets.uniform3d(3000, seed=9)
    k = 5
    plain = oracle.trueknn(pts, k, 0.02)
    same = oracle.trueknn_per_query(pts, k, np.full(len(pts), 0.02, np.float32))
    assert np.array_equal(same["idx"], plain["idx"]) and np.array_equal(same["dist"], plain["dist"])
    assert np.array_equal(same["intersections"], plain["intersections"]) and same["rounds"] == plain["rounds"]
    radii = np.where(np.arange(len(pts)) % 2 == 0, np.float32(0.01), np.float32(0.08)).astype(np.float32)
    mixed = oracle.trueknn_per_query(pts, k, radii)
    small, large = oracle.trueknn(pts, k, 0.01), oracle.trueknn(pts, k, float(np.float32(0.08)))
    even = np.arange(len(pts)) % 2 == 0
    assert np.array_equal(mixed["idx"][even], small["idx"][even]) and np.array_equal(mixed["idx"][~even], large["idx"][~even])
    assert np.array_equal(mixed["intersections"][~even], large["intersections"][~even])
