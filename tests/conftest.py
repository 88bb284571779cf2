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
