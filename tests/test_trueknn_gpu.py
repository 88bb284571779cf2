"""Parity of the HIP TrueKNN engine (through the C-ABI) against golden vectors and the CPU checker."""
import numpy as np
import pytest

import oracle
from owlraytracing_amd import _lib, datasets

from conftest import assert_rows_equal, assert_rows_match

pytestmark = pytest.mark.gpu

KERNELS = [_lib.KERNEL_LANE, _lib.KERNEL_WAVE, _lib.KERNEL_TEAM]
KERNEL_IDS = ["lane", "wave", "team"]


def _engine():
    from owlraytracing_amd.trueknn import TrueKNN
    return TrueKNN()


def _fb_view(fb_bytes, n, k):
    return fb_bytes.cpu().numpy().view(oracle.NEIGH_DTYPE).reshape(n, k)


@pytest.mark.parametrize("kernel", KERNELS, ids=KERNEL_IDS)
def test_golden_vectors(golden, kernel):
    eng = _engine()
    eng.build(golden["xyz"])
    k = int(golden["k"])
    if kernel != _lib.KERNEL_TEAM and k > 64:
        # the lane and wave kernels keep their lists in registers (k <= 64); larger k: the team walk with the lists in memory
        with pytest.raises(_lib.TknnError) as e:
            eng.solve(k, float(golden["start_radius"]), kernel=kernel)
        assert e.value.code == -5  # TKNN_E_UNSUPPORTED
        eng.close()
        return
    r = eng.solve(k, float(golden["start_radius"]), kernel=kernel, want_fb=True)
    # (the team kernel finishes packets whose candidate sets outgrow its lists with lane rounds or the
    # wave kernel and still reports itself)
    assert r["info"]["kernel_used"] in ((kernel, _lib.KERNEL_WAVE) if kernel == _lib.KERNEL_TEAM else (kernel,))
    assert r["info"]["rounds"] == int(golden["rounds"])
    assert np.float32(r["info"]["final_radius"]) == golden["final_radius"]
    # (crossroundties_*: exact distance ties between candidates of different rounds, which the replay --
    # and the engines' tie pass -- order by the round first seen, then by index)
    assert_rows_match(r["idx"].cpu().numpy(), r["dist"].cpu().numpy(), r["intersections"].cpu().numpy(), golden)
    if k > 64:
        assert r["info"]["list_capacity"] == (k + 15) // 16 * 16 and r["info"]["tie_rows"] == 0  # three-word keys: rows are final
        auto = eng.solve(k, float(golden["start_radius"]))  # TKNN_KERNEL_AUTO takes the same path
        assert np.array_equal(auto["idx"].cpu().numpy(), r["idx"].cpu().numpy())
    if golden["name"].startswith("crossroundties"):
        assert r["info"]["tie_rows"] > 0 and r["info"]["tie_rows_left"] == 0
    assert r["info"]["total_intersections"] == int(golden["intersections"].sum())
    # frameBuffer image = what the reference leaves behind (GeomTypes.h:22-28 records)
    ref = oracle.trueknn(golden["xyz"], k, float(golden["start_radius"]))
    fb = _fb_view(r["fb"], len(golden["xyz"]), k)
    want = ref["fb"].reshape(len(golden["xyz"]), k)
    for field in ("ind", "dist", "numNeighbors", "intersections"):
        assert np.array_equal(fb[field], want[field]), field
    eng.close()


@pytest.mark.parametrize("kernel", KERNELS, ids=KERNEL_IDS)
@pytest.mark.parametrize("n,k,seed", [(100_000, 5, 0), (50_000, 10, 1), (30_000, 16, 2), (20_000, 3, 3),
                                      (30_000, 17, 4), (40_000, 24, 5), (25_000, 32, 6), (25_000, 33, 7), (22_000, 40, 14), (20_000, 48, 15), (20_000, 49, 16), (20_000, 50, 8),
                                      (20_000, 64, 9), (12_000, 65, 10), (10_000, 100, 11), (6_000, 256, 12), (5_000, 1024, 13)])
def test_against_oracle_uniform(kernel, n, k, seed):
    xyz = datasets.uniform3d(n, seed=seed)
    r0 = datasets.start_radius(n, k)
    eng = _engine()
    eng.build(xyz)
    if k > 64 and kernel != _lib.KERNEL_TEAM:  # (k above the register lists: the team walk with the lists in memory only)
        with pytest.raises(_lib.TknnError) as e:
            eng.solve(k, r0, kernel=kernel)
        assert e.value.code == -5
        eng.close()
        return
    ref = oracle.trueknn(xyz, k, r0)
    r = eng.solve(k, r0, kernel=kernel)
    assert r["info"]["rounds"] == ref["rounds"]
    assert_rows_equal(r["idx"].cpu().numpy(), r["dist"].cpu().numpy(), ref["idx"], ref["dist"])
    assert r["info"]["tie_rows_left"] == 0
    assert np.array_equal(r["intersections"].cpu().numpy(), ref["intersections"])
    eng.close()


@pytest.mark.parametrize("kernel", KERNELS, ids=KERNEL_IDS)
def test_against_oracle_clustered_and_duplicates(kernel):
    xyz = datasets.gaussian_mixture3d(60_000, components=16, sigma=0.01, seed=5)
    xyz[::11] = xyz[3::11][: len(xyz[::11])]  # exact duplicates: ties at distance 0
    ref = oracle.trueknn(xyz, 8, 0.002)
    eng = _engine()
    eng.build(xyz)
    r = eng.solve(8, 0.002, kernel=kernel)
    assert r["info"]["rounds"] == ref["rounds"]
    assert np.array_equal(r["idx"].cpu().numpy(), ref["idx"])
    assert np.array_equal(r["dist"].cpu().numpy(), ref["dist"])
    assert np.array_equal(r["intersections"].cpu().numpy(), ref["intersections"])
    eng.close()


@pytest.mark.parametrize("kernel", KERNELS, ids=KERNEL_IDS)
def test_planar_input_and_heavy_tail(kernel):
    xy = datasets.taxi_like2d(40_000, components=32, seed=2)
    xyz = datasets.pad_to_3d(xy)
    ref = oracle.trueknn(xyz, 6, 0.0005)
    eng = _engine()
    eng.build(xy)  # 2-D input is padded with z = 0 like hostCode.cpp:115-118
    r = eng.solve(6, 0.0005, kernel=kernel)
    assert r["info"]["rounds"] == ref["rounds"]
    assert np.array_equal(r["idx"].cpu().numpy(), ref["idx"])
    assert np.array_equal(r["dist"].cpu().numpy(), ref["dist"])
    assert np.array_equal(r["intersections"].cpu().numpy(), ref["intersections"])
    eng.close()


@pytest.mark.parametrize("kernel", KERNELS, ids=KERNEL_IDS)
def test_points_on_box_faces_at_every_magnitude(kernel):
    """Satellites +-2 ulps around the box faces fl(c -+ r) of 400 anchors with coordinates from 1e-3
    to 600: the cases where bounds on |c - q| cannot decide the candidate test (team kernel's
    literal fallback) and where the exact thresholds of the lane/wave kernels sit."""
    xyz = datasets.boundary_band(400, 0.01, seed=11)
    ref = oracle.trueknn(xyz, 7, 0.01)
    eng = _engine()
    eng.build(xyz)
    r = eng.solve(7, 0.01, kernel=kernel)
    assert r["info"]["rounds"] == ref["rounds"]
    assert np.array_equal(r["intersections"].cpu().numpy(), ref["intersections"])
    assert np.array_equal(r["idx"].cpu().numpy(), ref["idx"])
    assert np.array_equal(r["dist"].cpu().numpy().view(np.int32), ref["dist"].view(np.int32))
    eng.close()


@pytest.mark.parametrize("kernel", KERNELS, ids=KERNEL_IDS)
def test_points_with_nan_coordinates_are_nobodys_candidates(kernel):
    """A NaN coordinate fails every closed-box comparison (deviceCode.cu:38-56 on IEEE floats), so
    such a point is never a candidate; as a query it never finishes (allow_unfinished reports it)."""
    xyz = datasets.uniform3d(20_000, seed=21)
    bad = np.array([5, 777, 19_999])
    xyz[bad[0], 0] = np.nan
    xyz[bad[1], 1] = np.nan
    xyz[bad[2]] = np.nan
    good = np.setdiff1d(np.arange(len(xyz)), bad)
    k, r0 = 5, datasets.start_radius(len(xyz), 5)
    ref = oracle.trueknn(xyz[good], k, r0)
    eng = _engine()
    eng.build(xyz)
    r = eng.solve(k, r0, kernel=kernel, max_rounds=ref["rounds"] + 2, allow_unfinished=True)
    assert r["info"]["unfinished"] == len(bad)
    idx = r["idx"].cpu().numpy()[good]
    assert not np.isin(idx, bad).any()
    assert np.array_equal(idx, good[ref["idx"]])
    assert np.array_equal(r["dist"].cpu().numpy()[good].view(np.int32), ref["dist"].view(np.int32))
    assert np.array_equal(r["intersections"].cpu().numpy()[good], ref["intersections"])
    eng.close()


@pytest.mark.parametrize("tail", ["walk", "lane", "wave"])
@pytest.mark.parametrize("k", [7, 24, 40])
def test_team_kernel_tails_agree_when_everything_is_handed_over(monkeypatch, tail, k):
    """A start radius far too large for the density of the cluster cores: nearly every packet outgrows the team kernel's LDS lists
    at level 0, so the rows come from its tail -- the team walk (one query per team, subtrees counted),
    the lane rounds or the wave kernel in subset mode.  Each must reproduce the checker."""
    monkeypatch.setenv("TKNN_TEAM_TAIL", tail)
    xyz = datasets.gaussian_mixture3d(40_000, components=3, sigma=0.004, seed=13)
    xyz[::7] = xyz[1::7][: len(xyz[::7])]  # duplicates
    ref = oracle.trueknn(xyz, k, 0.002)
    eng = _engine()
    eng.build(xyz)
    r = eng.solve(k, 0.002, kernel=_lib.KERNEL_TEAM)
    assert r["info"]["rounds"] == ref["rounds"] and ref["rounds"] >= 3  # the sparse fringe needs more levels
    assert np.array_equal(r["intersections"].cpu().numpy(), ref["intersections"])
    assert np.array_equal(r["idx"].cpu().numpy(), ref["idx"])
    assert np.array_equal(r["dist"].cpu().numpy().view(np.int32), ref["dist"].view(np.int32))
    assert r["info"]["total_intersections"] == int(ref["intersections"].sum())
    eng.close()


@pytest.mark.parametrize("kernel", KERNELS, ids=KERNEL_IDS)
def test_exact_distance_ties_across_rounds(kernel):
    """Candidates at bit-identical distances that entered the reference's persistent list in different
    rounds keep that order whatever their indices (oracle/trueknn_oracle.c, decision 2).  The kernels
    list by (dist, index) and flag such rows; tie_fix_kernel redoes them with the round as the second
    key word.  The set is built so that index order and round order disagree."""
    xyz = datasets.cross_round_ties()
    ref = oracle.trueknn(xyz, 2, 1.0)
    assert ref["rounds"] >= 2
    # the construction does produce rows where plain (dist, index) order is not the replay's
    plain = np.lexsort((ref["idx"], ref["dist"]), axis=1)
    assert (plain != np.arange(2)[None, :]).any()
    eng = _engine()
    eng.build(xyz)
    r = eng.solve(2, 1.0, kernel=kernel, want_fb=True)
    assert np.array_equal(r["intersections"].cpu().numpy(), ref["intersections"])
    assert_rows_equal(r["idx"].cpu().numpy(), r["dist"].cpu().numpy(), ref["idx"], ref["dist"])
    assert r["info"]["tie_rows"] > 0 and r["info"]["tie_rows_left"] == 0
    fb = _fb_view(r["fb"], len(xyz), 2)
    assert np.array_equal(fb["ind"], ref["fb"].reshape(len(xyz), 2)["ind"])
    # without distances to gate with, the pass still lands on the same rows
    r2 = eng.solve(2, 1.0, kernel=kernel, out={"dist": None})
    assert np.array_equal(r2["idx"].cpu().numpy(), ref["idx"])
    eng.close()


@pytest.mark.parametrize("kernel", KERNELS, ids=KERNEL_IDS)
def test_frame_buffer_only_solves_order_exact_ties_like_the_replay(kernel):
    """A call that asks for the reference's frameBuffer alone (d_idx = d_dist = d_intersections = NULL -- how the reference's
    own host code receives its results) still gets the tie pass: the records of the cross-round fixture and of a lattice
    (tens of thousands of exact-distance ties) equal the replay's, field for field."""
    for name, xyz, k, r0 in (("crossroundties", datasets.cross_round_ties(), 2, 1.0),
                             ("lattice3d", _lattice(14, 3, 6), 6, 0.02),
                             ("lattice2d_k16", _lattice(50, 2, 16), 16, 0.011)):
        ref = oracle.trueknn(xyz, k, r0)
        eng = _engine()
        eng.build(xyz)
        r = eng.solve(k, r0, kernel=kernel, fb_only=True)
        assert "idx" not in r and "dist" not in r and "intersections" not in r
        assert r["info"]["tie_rows_left"] == 0 and (r["info"]["tie_rows"] > 0 or name != "crossroundties"), name
        fb = _fb_view(r["fb"], len(xyz), k)
        want = ref["fb"].reshape(len(xyz), k)
        for field in ("ind", "dist", "numNeighbors", "intersections"):
            assert np.array_equal(fb[field], want[field]), (name, field)
        eng.close()


def _lattice(m, dims, seed, drop=0.2):
    g = np.arange(m, dtype=np.float32) / np.float32(32)
    if dims == 3:
        xyz = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    else:
        xy = np.stack(np.meshgrid(g, g, indexing="ij"), -1).reshape(-1, 2)
        xyz = np.concatenate([xy, np.zeros((len(xy), 1), np.float32)], 1)
    rng = np.random.default_rng(seed)
    xyz = xyz[rng.random(len(xyz)) > drop]
    return np.ascontiguousarray(xyz[rng.permutation(len(xyz))])


_TIE_SETS = {}


@pytest.mark.parametrize("kernel", KERNELS, ids=KERNEL_IDS)
@pytest.mark.parametrize("tail", [None, "walk", "lane"])
@pytest.mark.parametrize("k", [1, 5, 16, 17, 32, 33, 48, 64, 65, 100])
def test_tie_heavy_sets_equal_the_replay(kernel, tail, k, monkeypatch):
    """Lattices with holes and coarsely quantised coordinates: almost every row has bit-identical
    distances inside it or at its end, many of them between candidates of different rounds.  k = 16,
    32, 48, 64 fill a team's list to the last entry (the tie with the best candidate left out is then
    seen by what leaves the list, not by a spare entry)."""
    if tail and kernel != _lib.KERNEL_TEAM:
        pytest.skip("tails belong to the team kernel")
    if k > 64 and (tail or kernel != _lib.KERNEL_TEAM):
        pytest.skip("k > 64: the team walk with the lists in memory, no packet kernel, no tails")
    if tail:
        monkeypatch.setenv("TKNN_TEAM_TAIL", tail)
    if k not in _TIE_SETS:  # the replay of a set is shared by the kernels and tails
        _TIE_SETS.clear()
        sets = [(_lattice(14, 3, k), 0.02), (_lattice(50, 2, k), 0.011),
                ((np.round(datasets.uniform3d(20_000, seed=k) * 64) / 64).astype(np.float32), 0.01)]
        _TIE_SETS[k] = [(xyz, r0, oracle.trueknn(xyz, k, r0)) for xyz, r0 in sets]
    for xyz, r0, ref in _TIE_SETS[k]:
        eng = _engine()
        eng.build(xyz)
        r = eng.solve(k, r0, kernel=kernel)
        assert np.array_equal(r["intersections"].cpu().numpy(), ref["intersections"])
        assert_rows_equal(r["idx"].cpu().numpy(), r["dist"].cpu().numpy(), ref["idx"], ref["dist"])
        if k > 64:
            assert r["info"]["tie_rows"] == 0  # (the lists order by (distance, first level, index) from the start)
        else:
            assert r["info"]["tie_rows"] > 0 and r["info"]["tie_rows_left"] == 0
        eng.close()


@pytest.mark.parametrize("look", ["1", "0"])
@pytest.mark.parametrize("k", [5, 16, 24, 48])
def test_ties_between_duplicates_keep_their_rows(k, look, monkeypatch):
    """Duplicated points tie in every row that holds both copies, always as candidates of one level: the tie pass looks at
    the written row first and walks only for rows whose tied neighbours differ in level (or tie with a candidate left out).
    Sets: uniform points with a quarter of them copies; the taxi-like set (5 % copies, heavy tails); a lattice with holes --
    ties across levels, the walk's case -- with copies on top.  TKNN_TIE_LOOK=0 is the pass without the look: same rows."""
    monkeypatch.setenv("TKNN_TIE_LOOK", look)
    rng = np.random.default_rng(100 + k)
    uni = datasets.uniform3d(24_000, seed=k)
    uni[rng.choice(len(uni), 6000, replace=False)] = uni[rng.integers(0, len(uni), 6000)]
    lat = _lattice(22, 3, k)
    lat = np.ascontiguousarray(np.concatenate([lat, lat[rng.integers(0, len(lat), len(lat) // 5)]])[rng.permutation(len(lat) + len(lat) // 5)])
    sets = [(uni, datasets.start_radius(len(uni), k)), (datasets.pad_to_3d(datasets.taxi_like2d(30_000, components=12, seed=k)), 0.002),
            (lat, 0.02)]
    flagged = 0
    for xyz, r0 in sets:
        ref = oracle.trueknn(xyz, k, r0)
        eng = _engine()
        eng.build(xyz)
        r = eng.solve(k, r0, kernel=_lib.KERNEL_TEAM)
        assert np.array_equal(r["intersections"].cpu().numpy(), ref["intersections"])
        assert_rows_equal(r["idx"].cpu().numpy(), r["dist"].cpu().numpy(), ref["idx"], ref["dist"])
        assert r["info"]["tie_rows_left"] == 0
        flagged += r["info"]["tie_rows"]
        fb = _fb_view(eng.solve(k, r0, kernel=_lib.KERNEL_TEAM, fb_only=True)["fb"], len(xyz), k)  # (the look reads the records)
        want = ref["fb"].reshape(len(xyz), k)
        for field in ("ind", "dist"):
            assert np.array_equal(fb[field], want[field]), field
        eng.close()
    assert flagged > 100


def test_wave_kernel_redo_path_with_more_tie_rows_than_the_list_holds(monkeypatch):
    """The wave kernel re-solves its queries when its LDS stack overflows; TKNN_WAVE_FORCE_REDO takes that path without a
    pathological tree.  On a lattice nearly every row is flagged for the tie pass: more than the 4096 the device-side
    list holds, so the pass must take its row count from the flags, not from the list (the list's length had
    been read as the number of rows)."""
    xyz = _lattice(28, 3, 5)  # ~17 500 points, every one with ties
    k, r0 = 8, 0.02
    ref = oracle.trueknn(xyz, k, r0)
    monkeypatch.setenv("TKNN_WAVE_FORCE_REDO", "1")  # read when the engine is created
    eng = _engine()
    eng.build(xyz)
    r = eng.solve(k, r0, kernel=_lib.KERNEL_WAVE)
    assert r["info"]["tie_rows"] > 4096 and r["info"]["tie_rows_left"] == 0
    assert np.array_equal(r["intersections"].cpu().numpy(), ref["intersections"])
    assert_rows_equal(r["idx"].cpu().numpy(), r["dist"].cpu().numpy(), ref["idx"], ref["dist"])
    eng.close()


def test_candidate_thresholds_equal_the_literal_box_test():
    """lo <= c <= hi must select exactly the c with fl(c - r) <= q <= fl(c + r), including near
    zero, across binades and at exact box boundaries (where a per-ulp walk would take forever)."""
    import ctypes
    import torch
    rng = np.random.default_rng(0)
    q = np.concatenate([
        rng.random(4000, dtype=np.float32), np.float32([0, 0.125, 0.25, 1.375, 1e-30, -1e-30, 0.005, 0.0050001]),
        (rng.random(2000, dtype=np.float32) - 0.5) * np.float32(1e-3), rng.normal(0, 100, 1000).astype(np.float32)])
    r = np.concatenate([
        np.full(4000, 0.005, np.float32), np.float32([0.125, 0.125, 0.125, 0.125, 0.005, 0.005, 0.005, 0.005]),
        np.exp(rng.uniform(np.log(1e-6), np.log(10), 3000)).astype(np.float32)])
    # make some q exactly r-aligned so that q - r == 0 / boundaries coincide
    q[:500] = (np.arange(500) % 7).astype(np.float32) * r[:500]
    tq, tr = torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda()
    lo, hi = torch.empty_like(tq), torch.empty_like(tq)
    _lib.check(_lib.load().tknnDebugThresholds(
        ctypes.c_void_p(tq.data_ptr()), ctypes.c_void_p(tr.data_ptr()), len(q),
        ctypes.c_void_p(lo.data_ptr()), ctypes.c_void_p(hi.data_ptr()), None))
    lo, hi = lo.cpu().numpy(), hi.cpu().numpy()
    assert np.all(lo <= hi)
    def literal(c):
        return ((c - r).astype(np.float32) <= q) & (q <= (c + r).astype(np.float32))
    for base in (lo, hi):
        c = base.copy()
        for _ in range(3):  # walk a few ulps outward and inward around both thresholds
            for cand in (c, np.nextafter(c, np.float32(-np.inf)), np.nextafter(c, np.float32(np.inf))):
                assert np.array_equal((lo <= cand) & (cand <= hi), literal(cand))
            c = np.nextafter(c, np.float32(np.inf) if base is hi else np.float32(-np.inf))
    # interior and far-away points
    for cand in (q, q + r * np.float32(0.5), q - r * np.float32(3), q + r * np.float32(3)):
        cand = cand.astype(np.float32)
        assert np.array_equal((lo <= cand) & (cand <= hi), literal(cand))


@pytest.mark.parametrize("n", [2, 3, 64, 65, 129, 20_000])
def test_lbvh_side_tables(n):
    """The builder's two side tables (RT-DBSCAN climbs and starts its walks with them): split_owner gives every node's
    parent in O(1) -- an internal node is a left child iff it is the LAST position of its range --, and block_paths holds,
    per 64 sorted slots, the deepest internal node whose range holds the whole block and its four nearest ancestors (the
    root where the path is shorter).  Checked against the exported tree by a walk from the root."""
    xyz = datasets.gaussian_mixture3d(max(n, 8), components=4, sigma=0.02, seed=19)[:n].copy()
    if n > 300:
        xyz[100:300] = xyz[0]  # identical Morton codes
    eng = _engine()
    eng.build(xyz)
    t = eng.export_tree()
    eng.close()
    nodes = t["nodes"]
    split, other = nodes[:, 3].view(np.int32), nodes[:, 7].view(np.int32)
    i = np.arange(n - 1)
    first, last = np.minimum(i, other), np.maximum(i, other)
    owner = t["split_owner"]
    assert sorted(split.tolist()) == list(range(n - 1)) and np.array_equal(owner[split], i)  # every position splits one node
    # parents by the rule, against the children each node names
    left = np.where(first == split, -1, split)  # internal left child (or -1: a leaf)
    right = np.where(last == split + 1, -1, split + 1)
    parent = np.full(n - 1, -1)
    parent[left[left >= 0]] = i[left >= 0]
    parent[right[right >= 0]] = i[right >= 0]
    for node in range(1, n - 1):
        by_rule = owner[node] if node == last[node] else owner[node - 1]
        assert by_rule == parent[node], node
    paths = t["block_paths"]
    assert paths.shape == ((n + 63) // 64, 5)
    for b in range(paths.shape[0]):
        lo, hi = 64 * b, min(n - 1, 64 * b + 63)
        node, trail = 0, [0, 0, 0, 0]
        while True:  # the walk the table stands for
            if hi <= split[node] and first[node] != split[node]:
                nxt = split[node]
            elif lo > split[node] and last[node] != split[node] + 1:
                nxt = split[node] + 1
            else:
                break
            trail = trail[1:] + [node]
            node = nxt
        assert first[node] <= lo and hi <= last[node]
        assert paths[b].tolist() == trail + [node], b


def test_lbvh_invariants():
    xyz = datasets.gaussian_mixture3d(20_000, components=8, sigma=0.02, seed=9)
    xyz[100:200] = xyz[0]  # identical Morton codes: index tie-break in the radix tree
    eng = _engine()
    eng.build(xyz)
    t = eng.export_tree()
    n = len(xyz)
    nodes = t["nodes"]
    lo = nodes[:, 0:3].view(np.float32)
    hi = nodes[:, 4:7].view(np.float32)
    split = nodes[:, 3].view(np.int32)
    other = nodes[:, 7].view(np.int32)
    prim = t["prim_id"]
    assert sorted(prim.tolist()) == list(range(n))
    pts = xyz[prim]
    i = np.arange(n - 1)
    first, last = np.minimum(i, other), np.maximum(i, other)
    assert first[0] == 0 and last[0] == n - 1
    assert np.all((split >= first) & (split < last))
    # boxes are exactly the bounds of the covered sorted range
    for node in list(range(0, 200)) + list(range(n - 200, n - 1)):
        seg = pts[first[node]:last[node] + 1]
        assert np.array_equal(lo[node], seg.min(0)) and np.array_equal(hi[node], seg.max(0))
    # a rope-guided walk that never descends visits the root only; one that always descends
    # visits every leaf exactly once, in sorted order
    ref, seen, steps = 0, [], 0
    while ref != np.int32(-2**31) and steps < 3 * n:
        steps += 1
        if ref >= 0:
            ref = (~split[ref]) if first[ref] == split[ref] else split[ref]
        else:
            seen.append(~ref)
            ref = t["rope_leaf"][~ref]
    assert seen == list(range(n))
    assert t["rope_node"][0] == np.int32(-2**31)
    eng.close()


def test_argument_errors_are_reported_not_hidden():
    eng = _engine()
    with pytest.raises(_lib.TknnError) as e:
        eng.solve(3, 0.1)
    assert e.value.code == -3  # TKNN_E_STATE
    xyz = datasets.uniform3d(10, seed=0)
    eng.build(xyz)
    for k, r0, code in ((10, 0.1, -1), (0, 0.1, -1), (3, 0.0, -1), (3, float("inf"), -1), (65, 0.1, -1), (1025, 0.1, -5)):
        with pytest.raises(_lib.TknnError) as e:
            eng.solve(k, r0)
        assert e.value.code == code
    # far-apart duplicates of fewer than k+1 distinct places still terminate; max_rounds is honoured
    with pytest.raises(_lib.TknnError) as e:
        eng.solve(3, 1e-30, max_rounds=3)
    assert e.value.code == -4
    # ... and capped at 127 (the tie flags keep the finishing level in seven bits): a start radius that needs 100 doublings is
    # served, and asking for a million rounds is the same as asking for 127
    far = eng.solve(3, 1e-30, max_rounds=1_000_000, kernel=_lib.KERNEL_TEAM)
    assert 90 <= far["info"]["rounds"] <= 127 and far["info"]["unfinished"] == 0
    ref = oracle.trueknn(xyz, 3, 1e-30, max_rounds=127)
    assert_rows_equal(far["idx"].cpu().numpy(), far["dist"].cpu().numpy(), ref["idx"], ref["dist"])
    eng.close()


@pytest.mark.parametrize("kernel", KERNELS, ids=KERNEL_IDS)
def test_full_size_properties_c2(kernel):
    """BASELINE config 2 (10 M points, k=10): size-independent properties + sampled oracle rows."""
    n, k = 10_000_000, 10
    xyz = datasets.uniform3d(n, seed=0)
    r0 = datasets.start_radius(n, k)
    eng = _engine()
    eng.build(xyz)
    r = eng.solve(k, r0, kernel=kernel)
    idx, dist, isect = r["idx"], r["dist"], r["intersections"]
    import torch
    ar = torch.arange(n, device=idx.device, dtype=torch.int32)[:, None]
    assert bool((idx != ar).all()) and bool((idx >= 0).all()) and bool((idx < n).all())
    assert bool((dist[:, 1:] >= dist[:, :-1]).all())
    assert int(isect.sum()) == r["info"]["total_intersections"]
    # checksum of checksums: recompute every distance from the indices on the GPU in fp64
    pts = torch.from_numpy(xyz).to(idx.device)
    d64 = (pts[idx.long()].double() - pts[:, None, :].double()).norm(dim=2)
    assert float((d64 - dist.double()).abs().max()) < 1e-6
    # sampled rows against the CPU checker (bit-exact)
    q = np.arange(0, n, 5003, dtype=np.int32)
    ref = oracle.trueknn(xyz, k, r0, query_ids=q)
    assert np.array_equal(idx[q.astype(np.int64)].cpu().numpy(), ref["idx"][q])
    assert np.array_equal(dist[q.astype(np.int64)].cpu().numpy(), ref["dist"][q])
    assert np.array_equal(isect[q.astype(np.int64)].cpu().numpy(), ref["intersections"][q])
    eng.close()


def test_full_size_c2_set_above_the_register_lists():
    """BASELINE config 2's point set at k = 100 (the reference takes any k, hostCode.cpp:111): the team walk with the lists in
    memory over all 10 M queries -- structure of all 10^9 list entries, every distance recomputed in fp64 (in slices), and the
    rows of every 5 003rd query bit-exact against the CPU replay."""
    import torch
    n, k = 10_000_000, 100
    xyz = datasets.uniform3d(n, seed=0)
    r0 = datasets.start_radius(n, k)
    eng = _engine()
    eng.build(xyz)
    r = eng.solve(k, r0)
    idx, dist, isect, info = r["idx"], r["dist"], r["intersections"], r["info"]
    assert info["unfinished"] == 0 and info["list_capacity"] == 112 and info["tie_rows"] == 0
    assert int(isect.sum()) == info["total_intersections"]
    pts = torch.from_numpy(xyz).to(idx.device)
    worst = 0.0
    step = 1_000_000
    for lo in range(0, n, step):
        i, d = idx[lo:lo + step], dist[lo:lo + step]
        ar = torch.arange(lo, lo + len(i), device=i.device, dtype=torch.int32)[:, None]
        assert bool((i != ar).all()) and bool((i >= 0).all()) and bool((i < n).all())
        assert bool((d[:, 1:] >= d[:, :-1]).all())
        d64 = (pts[i.long()].double() - pts[lo:lo + len(i), None, :].double()).norm(dim=2)
        worst = max(worst, float((d64 - d.double()).abs().max()))
        del d64, ar
    assert worst < 1e-6
    q = np.arange(0, n, 5003, dtype=np.int32)
    ref = oracle.trueknn(xyz, k, r0, query_ids=q)
    assert info["rounds"] == ref["rounds"]
    assert np.array_equal(idx[q.astype(np.int64)].cpu().numpy(), ref["idx"][q])
    assert np.array_equal(dist[q.astype(np.int64)].cpu().numpy(), ref["dist"][q])
    assert np.array_equal(isect[q.astype(np.int64)].cpu().numpy(), ref["intersections"][q])
    eng.close()


def test_full_size_config4_set_on_one_gpu():
    """BASELINE config 4's point set -- 100 M counter-based uniform points, k = 10 -- on ONE GPU (the run README / DESIGN
    quote a time for; the 8-GPU tiling of the same set is tests/test_distributed.py's business):
      * structure of all 10^8 rows: no self, indices in range, distances ascending, intersection counts add up;
      * every one of the 10^9 distances recomputed in fp64 from the indices;
      * 1 001 sampled rows bit-exact against the CPU checker (oracle.trueknn_rows: the replay of the reference's loop for
        the sampled queries over all 10^8 points), with their intersection counts and the number of rounds."""
    import torch
    n, k = 100_000_000, 10
    xyz = datasets.uniform3d_counter(0, n, seed=0)
    r0 = datasets.start_radius(n, k)
    eng = _engine()
    pts = torch.from_numpy(xyz).cuda()
    b = eng.build(pts)
    r = eng.solve(k, r0)
    idx, dist, isect, info = r["idx"], r["dist"], r["intersections"], r["info"]
    assert info["unfinished"] == 0 and info["tie_rows_left"] == 0
    assert int(isect.sum()) == info["total_intersections"]
    worst = 0.0
    step = 10_000_000
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        i, d = idx[lo:hi], dist[lo:hi]
        ar = torch.arange(lo, hi, device=i.device, dtype=torch.int32)[:, None]
        assert bool((i != ar).all()) and bool((i >= 0).all()) and bool((i < n).all())
        assert bool((d[:, 1:] >= d[:, :-1]).all())
        d64 = (pts[i.long()].double() - pts[lo:hi, None, :].double()).norm(dim=2)
        worst = max(worst, float((d64 - d.double()).abs().max()))
        del d64, ar
    assert worst < 1e-6
    q = np.arange(0, n, 99_901, dtype=np.int32)
    assert len(q) >= 1000
    ql = torch.from_numpy(q.astype(np.int64)).cuda()
    got_idx, got_dist, got_isect = idx[ql].cpu().numpy(), dist[ql].cpu().numpy(), isect[ql].cpu().numpy()
    del idx, dist, isect, r, pts
    eng.close()
    torch.cuda.empty_cache()
    ref = oracle.trueknn_rows(xyz, k, r0, q)
    assert np.array_equal(got_idx, ref["idx"]) and np.array_equal(got_dist, ref["dist"])
    assert np.array_equal(got_isect, ref["intersections"])
    print("config-4 set on one GPU: solve %.1f ms (kernel %.1f ms), build %.1f ms, %d rounds, %.1f intersection-program calls per query" % (
        info["solve_ms"], info["dominant_kernel_ms"], b["build_ms"], info["rounds"], info["total_intersections"] / n))


def test_exact_repair_gives_bruteforce_knn():
    """opt-in post-pass: rows with d_k beyond their box become exact kNN, others stay"""
    for n, k, seed in ((30_000, 10, 4), (8_000, 5, 5)):
        xyz = datasets.uniform3d(n, seed=seed) if seed == 4 else datasets.gaussian_mixture3d(n, 6, 0.04, seed)
        r0 = datasets.start_radius(n, k)
        eng = _engine()
        eng.build(xyz)
        r = eng.solve(k, r0, want_levels=True)
        before = r["idx"].clone()
        fixed = eng.repair_exact(r, k, r0)
        bi, bd = oracle.bruteforce_knn(xyz, k)
        assert np.array_equal(r["idx"].cpu().numpy(), bi)
        assert np.array_equal(r["dist"].cpu().numpy(), bd)
        changed = int((before != r["idx"]).any(dim=1).sum())
        assert 0 < changed <= fixed < n  # a sizeable share of reference rows is not exact kNN
        eng.close()


@pytest.mark.gpu
def test_per_query_radius_schedule():
    """tknnSolveOptions.d_start_radii (opt-in): every query doubles from a start radius of its own.  Rows,
    intersection counts and the number of rounds equal the checker's per-query statement (each query run through the
    reference's loop from its own radius): random radius classes on uniform points, radii from the local density on a
    clustered set (fewer rounds than one global radius needs), and a set full of exact-distance ties."""
    import torch

    from owlraytracing_amd import _lib
    from owlraytracing_amd.trueknn import TrueKNN
    eng = TrueKNN()
    rng = np.random.default_rng(77)
    cases = []
    pts = datasets.uniform3d(60_000, seed=41)
    cases.append(("uniform, four random classes", pts, 10, rng.choice(np.float32([0.004, 0.008, 0.016, 0.05]), len(pts))))
    pts = datasets.gaussian_mixture3d(50_000, components=7, sigma=0.03, seed=42)
    # a density guess per point: distance to the 3rd nearest of a thinned copy, clipped and rounded to a few values
    from scipy.spatial import cKDTree
    d3 = cKDTree(pts[::8].astype(np.float64)).query(pts.astype(np.float64), k=3)[0][:, 2]
    radii = np.float32(0.002) * np.float32(2.0) ** np.clip(np.round(np.log2(np.maximum(d3, 1e-6) / 0.002)), 0, 5).astype(np.float32)
    cases.append(("clustered, radii from the local density", pts, 8, radii.astype(np.float32)))
    lattice = np.stack(np.meshgrid(*[np.arange(14, dtype=np.float32) * np.float32(0.125)] * 3, indexing="ij"), -1).reshape(-1, 3)
    cases.append(("lattice (ties), two classes", lattice, 6, np.where(np.arange(len(lattice)) % 3 == 0, np.float32(0.07), np.float32(0.13)).astype(np.float32)))
    # ... and above the register lists (k > 64: the team walk with the lists in memory keeps the schedule per query, too)
    pts = datasets.uniform3d(8_000, seed=43)
    cases.append(("uniform, k = 70, three classes", pts, 70, rng.choice(np.float32([0.03, 0.06, 0.2]), len(pts))))
    cases.append(("lattice (ties), k = 80", lattice, 80, np.where(np.arange(len(lattice)) % 3 == 0, np.float32(0.2), np.float32(0.3)).astype(np.float32)))
    for name, pts, k, radii in cases:
        ref = oracle.trueknn_per_query(pts, k, radii)
        eng.build(pts)
        got = eng.solve(k, 1.0, start_radii=torch.from_numpy(radii), kernel=_lib.KERNEL_TEAM)
        assert_rows_equal(got["idx"].cpu().numpy(), got["dist"].cpu().numpy(), ref["idx"], ref["dist"])
        assert np.array_equal(got["intersections"].cpu().numpy(), ref["intersections"]), name
        assert got["info"]["rounds"] == ref["rounds"], name
        auto = eng.solve(k, 1.0, start_radii=torch.from_numpy(radii))  # AUTO resolves to the team kernels
        assert np.array_equal(auto["idx"].cpu().numpy(), got["idx"].cpu().numpy())
    # one global radius is the special case of equal radii
    pts = cases[0][1]
    eng.build(pts)
    a = eng.solve(10, 0.01)
    b = eng.solve(10, 123.0, start_radii=torch.full((len(pts),), 0.01))
    assert np.array_equal(a["idx"].cpu().numpy(), b["idx"].cpu().numpy()) and np.array_equal(a["dist"].cpu().numpy(), b["dist"].cpu().numpy())
    assert np.array_equal(a["intersections"].cpu().numpy(), b["intersections"].cpu().numpy())
    # not served by the other kernels
    for kern in (_lib.KERNEL_LANE, _lib.KERNEL_WAVE):
        with pytest.raises(_lib.TknnError):
            eng.solve(10, 0.01, start_radii=torch.full((len(pts),), 0.01), kernel=kern)
    eng.close()
