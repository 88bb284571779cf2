#!/usr/bin/env python3
"""GPU box: per-kernel times of RT-DBSCAN on 50 M heavy-tailed 2-D points (BASELINE config 5's kind of set) at a few eps.
    python scripts/taxi_probe.py [eps ...]"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from owlraytracing_amd import datasets
from owlraytracing_amd.trueknn import TrueKNN
n = 50_000_000
pts = torch.from_numpy(datasets.pad_to_3d(datasets.taxi_like2d(n, components=40, seed=5))).cuda()
eng = TrueKNN(); eng.build(pts)
for eps in (0.0004, 0.0008, 0.002) if len(sys.argv) < 2 else [float(x) for x in sys.argv[1:]]:
    for _ in range(2):
        r = eng.dbscan(eps, 4)
    i = r["info"]
    print("eps", eps, "ms", round(i["solve_ms"],2), "core", round(i["core_ms"],2), "union", round(i["union_ms"],2), "label", round(i["label_ms"],2), "groups", i["groups"], "clusters", i["clusters"], "nodes", i["node_tests"], "union nodes", i["union_node_tests"], "points", i["point_tests"], flush=True)
