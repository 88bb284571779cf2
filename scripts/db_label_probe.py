#!/usr/bin/env python3
"""GPU box: how many points of BASELINE config 3 are not core, and what the label pass's walks cost (work counters)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from owlraytracing_amd import datasets  # noqa: E402
from owlraytracing_amd.trueknn import TrueKNN  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
pts = torch.from_numpy(datasets.gaussian_mixture3d(n, components=64, sigma=0.02, seed=1)).cuda()
eng = TrueKNN()
eng.build(pts)
r = eng.dbscan(0.01, 4)
core, labels, i = r["core"], r["labels"], r["info"]
not_core = int((core == 0).sum())
print("not core %d (%.3f %%), of them noise %d; label pass %.3f ms, its point tests %d; node tests: all %d, unions %d" % (
    not_core, 100.0 * not_core / n, int((labels < 0).sum()), i["label_ms"], i["label_point_tests"], i["node_tests"], i["union_node_tests"]))
