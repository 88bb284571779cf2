#!/usr/bin/env python3
"""GPU box: per-phase device times of tknnDbscan at BASELINE config 3 (best of a few calls) -- scripts/db_times.py [n] [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from owlraytracing_amd import datasets  # noqa: E402
from owlraytracing_amd.trueknn import TrueKNN  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
pts = torch.from_numpy(datasets.gaussian_mixture3d(n, components=64, sigma=0.02, seed=1)).cuda()
eng = TrueKNN()
eng.build(pts)
best = None
for _ in range(reps):
    i = eng.dbscan(0.01, 4)["info"]
    if best is None or i["solve_ms"] < best["solve_ms"]:
        best = i
print("lib %s: clusters %d  whole %.2f ms  core %.2f  union %.2f  label %.2f  (nodes %d points %d; unions: nodes %d points %d)" % (
    os.path.basename(os.environ.get("OWL_MI355X_LIB", "default")), best["clusters"], best["solve_ms"], best["core_ms"], best["union_ms"], best["label_ms"],
    best["node_tests"], best["point_tests"], best.get("union_node_tests", -1), best.get("union_point_tests", -1)), flush=True)
