#!/usr/bin/env python3
"""GPU box: the group-union kernel's two rarely taken paths against the CPU spec (test infrastructure: uses oracle/).
  1. depth-first popping: a library built with -DTKNN_DB_STACK=320 (OWL_MI355X_LIB=...libowl_mi355x_s320.so)
  2. the host's fallback after a stack overflow: the diagnostic library with TKNN_DB_DIAG=16
Usage: OWL_MI355X_LIB=<lib> [TKNN_DB_DIAG=16] python scripts/db_fallback_check.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from owlraytracing_amd import datasets  # noqa: E402
from owlraytracing_amd.trueknn import TrueKNN  # noqa: E402

eng = TrueKNN()
for name, xyz, eps, min_pts in [
    ("gmm", datasets.gaussian_mixture3d(300_000, components=16, sigma=0.02, seed=2), 0.01, 4),
    ("taxi2d", datasets.pad_to_3d(datasets.taxi_like2d(200_000, components=20, seed=3)), 0.002, 5),
    ("uniform", datasets.uniform3d(100_000, seed=4), 0.02, 4),
]:
    eps = float(np.float32(eps))
    ref = oracle.dbscan(xyz, eps, min_pts)
    eng.build(xyz)
    got = eng.dbscan(eps, min_pts)
    assert np.array_equal(got["labels"].cpu().numpy(), ref["labels"]), name
    assert np.array_equal(got["core"].cpu().numpy(), ref["core"]), name
    assert got["info"]["clusters"] == ref["clusters"], name
    print("ok", name, "clusters", ref["clusters"], "groups", got["info"]["groups"], "union launches", got["info"]["union_launches"], flush=True)
print("library:", os.environ.get("OWL_MI355X_LIB", "default"), "TKNN_DB_DIAG =", os.environ.get("TKNN_DB_DIAG", "-"))
