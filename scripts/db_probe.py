#!/usr/bin/env python3
"""RT-DBSCAN at BASELINE config 3 under a few TKNN_DB_DIAG / chunk / grid / split settings (GPU box; measurements only).
TKNN_DB_DIAG needs the diagnostic library: make -C owlraytracing_amd/csrc DIAG=1, OWL_MI355X_LIB=.../libowl_mi355x_diag.so."""
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
for spec in sys.argv[2:] or ["0"]:  # diag[:chunk[:grid[:split]]]
    diag, chunk, grid, split = (spec.split(":") + ["", "", ""])[:4]
    if split:
        os.environ["TKNN_DB_SPLIT"] = split
    os.environ["TKNN_DB_DIAG"] = diag
    if chunk:
        os.environ["TKNN_DB_CHUNK"] = chunk
    if grid:
        os.environ["TKNN_DB_GRID"] = grid
    for _ in range(2):
        r = eng.dbscan(0.01, 4)
    i = r["info"]
    print("diag", spec, "clusters", i["clusters"], "ms", round(i["solve_ms"], 2), "core", round(i["core_ms"], 2), "union", round(i["union_ms"], 2),
          "label", round(i["label_ms"], 2), "nodes", i["node_tests"], "points", i["point_tests"], flush=True)
