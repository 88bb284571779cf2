"""tools/compare_reference_dump.py: the verdict an OptiX-side dump of the reference would get.

The reference's own neighbour print (samples/s01-trueknn/hostCode.cpp:312-321, commented out there)
cannot be produced in this repository, so the dump is synthesised from the CPU checker's rows in the
shapes a real one can have: partial lists of earlier rounds before the final rows, distances moved by a
couple of ulps (fast-math), exact-distance ties in another order, a candidate on a box face decided the
other way, six-digit default precision -- and one row that is simply wrong."""
import io
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
from owlraytracing_amd import datasets

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import compare_reference_dump as crd  # noqa: E402


def _dump_text(idx, dist, fmt="%.9g", partial_rounds=1, with_point_lines=True, pts=None):
    """What hostCode.cpp:312-321 writes: every round prints rows from query 0 up to the first unfinished one."""
    n, k = idx.shape
    out = io.StringIO()
    rng = np.random.default_rng(3)
    for _ in range(partial_rounds):  # earlier rounds: stale, half-filled lists up to some query
        stop = int(rng.integers(1, n))
        for j in range(stop):
            if with_point_lines and pts is not None:
                out.write("Point %d: (%g, %g, %g)\n" % (j, pts[j, 0], pts[j, 1], pts[j, 2]))
            for i in range(k):
                out.write("%d,%d,%s\n" % (j, -1 if i else int(idx[j, 0]), "3.40282e+38" if i else fmt % dist[j, 0]))
    for j in range(n):
        if with_point_lines and pts is not None:
            out.write("Point %d: (%g, %g, %g)\n" % (j, pts[j, 0], pts[j, 1], pts[j, 2]))
        for i in range(k):
            out.write("%d,%d,%s\n" % (j, int(idx[j, i]), fmt % dist[j, i]))
    return out.getvalue()


def _ulp_shift(x, by):
    return (np.asarray(x, np.float32).view(np.int32) + by).view(np.float32)


def test_parse_takes_the_final_row_of_every_query():
    pts = datasets.uniform3d(500, seed=11)
    ref = oracle.trueknn(pts, 4, 0.02)
    text = _dump_text(ref["idx"], ref["dist"], partial_rounds=3, pts=pts)
    idx, dist, tokens, seen = crd.parse_dump(io.StringIO(text), 500, 4)
    assert seen.all()
    assert np.array_equal(idx, ref["idx"])
    assert np.array_equal(dist.astype(np.float32).view(np.int32), ref["dist"].view(np.int32))  # %.9g round-trips fp32
    assert crd.significant_digits(tokens.ravel().tolist()) == 9


def test_verdicts_on_a_synthetic_dump():
    # duplicates give exact-distance ties; the rest is uniform
    n, k, r0 = 3000, 5, 0.02
    pts = datasets.uniform3d(n, seed=12)
    pts[100:110] = pts[50]          # ten copies of one point: rows full of zero-distance ties
    pts[200] = pts[201] = pts[202]  # a triple
    ref = oracle.trueknn(pts, k, r0)
    idx, dist = ref["idx"].astype(np.int64).copy(), ref["dist"].copy()
    # level of a row = first level whose closed box holds k others (the checker's definition)
    levels = np.zeros(n, np.int32)
    for q in range(n):
        l, r = 0, np.float32(r0)
        while True:
            inside = np.all((pts - r <= pts[q]) & (pts[q] <= pts + r), axis=1)
            if inside.sum() - 1 >= k:
                break
            l, r = l + 1, np.float32(r * np.float32(2))
        levels[q] = l
    want = {}
    # (1) fast-math: a few distances off by one or two ulps
    for q in (5, 6, 7):
        dist[q, 2] = _ulp_shift(dist[q, 2], 2)
        want[q] = "distance"
    # (2) exact ties printed in another order
    q = 50
    run = np.nonzero(dist[q] == dist[q, 0])[0]
    assert len(run) >= 2
    idx[q, run] = idx[q, run][::-1]
    want[q] = "tie-order"
    # (3) a wrong row
    idx[9] = (idx[9] + 17) % n
    want[9] = "mismatch"
    text = _dump_text(idx, dist, partial_rounds=2, pts=pts)
    p_idx, p_dist, tokens, seen = crd.parse_dump(io.StringIO(text), n, k)
    res = crd.classify(pts, k, r0, p_idx, p_dist, 9, ref["idx"], ref["dist"], levels)
    for q, v in want.items():
        assert res["verdict"][q] == v, (q, res["verdict"][q], v)
    assert res["counts"]["identical"] == n - len(want)
    assert res["counts"]["mismatch"] == 1
    assert res["worst_distance_ulps"] == 2
    # with two ulps of slack the fast-math rows count as identical
    res2 = crd.classify(pts, k, r0, p_idx, p_dist, 9, ref["idx"], ref["dist"], levels, ulps=2)
    assert res2["counts"]["distance"] == 0 and res2["counts"]["identical"] == n - 2


def test_default_six_digit_precision_is_compared_at_that_precision():
    n, k, r0 = 800, 3, 0.03
    pts = datasets.uniform3d(n, seed=13)
    ref = oracle.trueknn(pts, k, r0)
    text = _dump_text(ref["idx"], ref["dist"], fmt="%g", pts=pts)  # what operator<<(float) prints
    p_idx, p_dist, tokens, seen = crd.parse_dump(io.StringIO(text), n, k)
    digits = crd.significant_digits(tokens.ravel().tolist())
    assert digits <= 6
    res = crd.classify(pts, k, r0, p_idx, p_dist, digits, ref["idx"], ref["dist"], None)
    assert res["counts"]["identical"] == n


def test_a_candidate_on_a_box_face_decided_the_other_way_is_recognised():
    """Query 0 at the origin, k = 2, r0 = 1: two certain candidates and one EXACTLY on the face x = 1 that is
    nearer than one of them.  The closed fp32 box takes it (this repo's decision 1); a run that does not
    lists the other point instead.  And a run in which the face candidate decides the LEVEL: with it the box
    of level 0 holds k others, without it the row comes from level 1."""
    pts = np.array([[0.0, 0.0, 0.0],
                    [1.0, 0.0, 0.0],      # on the face of the level-0 box, distance 1
                    [0.9, 0.9, 0.0],      # inside, distance 1.27
                    [0.5, 0.0, 0.0],      # inside, distance 0.5
                    [30.0, 30.0, 30.0], [31.0, 30.0, 30.0], [30.0, 31.0, 30.0]], np.float32)
    k, r0 = 2, 1.0
    ref = oracle.trueknn(pts, k, r0)
    assert ref["idx"][0].tolist() == [3, 1]
    alt_idx = ref["idx"].astype(np.int64).copy()
    alt_dist = ref["dist"].copy()
    alt_idx[0] = [3, 2]
    alt_dist[0] = [0.5, np.sqrt(np.float32(0.81) + np.float32(0.81), dtype=np.float32)]
    levels = np.zeros(len(pts), np.int32)
    levels[4:] = 0
    res = crd.classify(pts, k, r0, alt_idx, alt_dist.astype(np.float64), 9, ref["idx"], ref["dist"], levels, rows=[0])
    assert res["verdict"][0] == "box-face"
    # the same row with a neighbour that no face decision can produce is a mismatch
    alt_idx[0] = [3, 4]
    res = crd.classify(pts, k, r0, alt_idx, alt_dist.astype(np.float64), 9, ref["idx"], ref["dist"], levels, rows=[0])
    assert res["verdict"][0] == "mismatch"
    # level decided by the face candidate: k = 3 -> with point 1 the level-0 box holds 3 others (rows 3,1,2);
    # without it the run goes to level 1 (r = 2), whose box holds the same three: same row, fine -- so move
    # point 2 out to 1.5 (outside level 0, inside level 1) and add a nearer level-1 candidate
    pts2 = pts.copy()
    pts2[2] = [0.6, 0.6, 0.0]                         # inside level 0, distance 0.85
    pts2 = np.concatenate([pts2, np.array([[1.2, 0.0, 0.0]], np.float32)])  # only in the level-1 box, distance 1.2
    k = 3
    ref2 = oracle.trueknn(pts2, k, r0)
    assert sorted(ref2["idx"][0].tolist()) == [1, 2, 3]  # finishes at level 0 thanks to the face candidate
    alt = ref2["idx"].astype(np.int64).copy()
    altd = ref2["dist"].astype(np.float64).copy()
    alt[0] = [3, 2, 7]  # the run that rejected point 1 at level 0 and took it at level 1: nearest three of {1,2,3,7} ... = 3, 2, 1
    lv = np.zeros(len(pts2), np.int32)
    res = crd.classify(pts2, k, r0, alt, altd, 9, ref2["idx"], ref2["dist"], lv, rows=[0])
    # {3, 2, 7} is NOT explainable: at level 1 point 1 (distance 1.0) is nearer than 7 (1.2) and certain there
    assert res["verdict"][0] == "mismatch"


def test_command_line_with_precomputed_rows(tmp_path):
    n, k, r0 = 600, 4, 0.03
    pts = datasets.uniform3d(n, seed=14)
    ref = oracle.trueknn(pts, k, r0)
    csv = tmp_path / "pts.csv"
    datasets.write_csv_points(str(csv), pts)
    # the CSV is the input of both sides: rows are those of the points as the loader reads them back
    back = datasets.pad_to_3d(datasets.read_csv_points(str(csv), n, 3))
    ref = oracle.trueknn(back, k, r0)
    dump = tmp_path / "dump.txt"
    dump.write_text(_dump_text(ref["idx"], ref["dist"], pts=back))
    rows = tmp_path / "rows.npz"
    np.savez(rows, idx=ref["idx"], dist=ref["dist"])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "compare_reference_dump.py"), str(csv), str(n), "3",
                        repr(r0), str(k), str(dump), "--rows", str(rows)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert '"parity": "pinned"' in r.stdout and '"mismatch": 0' in r.stdout
