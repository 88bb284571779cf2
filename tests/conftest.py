import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0]
                  for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_golden(name):
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(params=golden_names())
def golden(request):
    g = load_golden(request.param)
    g["name"] = request.param
    return g


def assert_rows_match(idx, dist, isect, ref, exact_ties=True, rows=None):
    """Compare a solver's rows with a golden / oracle result.

    Distances and intersection counts must be bit-identical.  Indices must be identical when the
    solver promises the canonical (dist, index) order (``exact_ties``); otherwise only on rows
    whose content does not depend on candidate visit order, and as equal index *sets* where the
    k-th distance is strictly below every excluded candidate (tie-free rows).
    """
    r_idx, r_dist, r_isect = ref["idx"], ref["dist"], ref["intersections"]
    if rows is not None:
        r_idx, r_dist, r_isect = r_idx[rows], r_dist[rows], r_isect[rows]
    assert np.array_equal(np.asarray(dist).view(np.int32), r_dist.view(np.int32)), "distances differ"
    if isect is not None:
        assert np.array_equal(np.asarray(isect), r_isect), "intersection counts differ"
    if exact_ties:
        assert np.array_equal(np.asarray(idx), r_idx), "indices differ"
    else:
        free = ref["order_free"] if rows is None else ref["order_free"][rows]
        assert np.array_equal(np.asarray(idx)[free], r_idx[free]), "indices differ on order-free rows"


def assert_rows_equal(idx, dist, ref_idx, ref_dist):
    """Engine rows against the call-by-call replay (oracle.trueknn): distances bit-identical, indices
    identical -- including the order of candidates at bit-identical fp32 distances, which the replay (like
    the reference, whose lists persist over rounds: deviceCode.cu:77-85,116-134) orders by the round in
    which each first was a candidate and then by index.  On a mismatch says whether the differing rows
    differ only inside runs of equal distances (the tie pass) or elsewhere (a kernel)."""
    idx, dist, ref_idx, ref_dist = np.asarray(idx), np.asarray(dist), np.asarray(ref_idx), np.asarray(ref_dist)
    assert np.array_equal(dist.view(np.int32), ref_dist.view(np.int32)), "distances differ"
    rows = np.nonzero((idx != ref_idx).any(axis=1))[0]
    if len(rows) == 0:
        return
    outside = 0
    for q in rows[:1000]:
        d = dist[q].view(np.int32)
        for pos in np.nonzero(idx[q] != ref_idx[q])[0]:
            run = np.nonzero(d == d[pos])[0]
            if not ((len(run) >= 2 and sorted(idx[q][run]) == sorted(ref_idx[q][run])) or d[pos] == d[-1]):
                outside += 1
    raise AssertionError("%d rows differ in their indices (first %d; of the first 1000, %d positions outside exact-distance ties)"
                         % (len(rows), rows[0], outside))
