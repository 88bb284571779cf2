#!/usr/bin/env python3
"""GPU box: tknnDbscanAuto on BASELINE config 5's point set (or a smaller one): scripts/db_auto_times.py [n] [max_noise]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from owlraytracing_amd import datasets  # noqa: E402
from owlraytracing_amd.trueknn import TrueKNN  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
max_noise = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
pts = torch.from_numpy(datasets.pad_to_3d(datasets.taxi_like2d(n, components=256, seed=2))).cuda()
eng = TrueKNN()
b = eng.build(pts)
for _ in range(3):
    torch.cuda.synchronize()
    t = time.perf_counter()
    i = eng.dbscan_auto(0.00005, 4, max_noise)["info"]
    torch.cuda.synchronize()
    w = (time.perf_counter() - t) * 1e3
print("lib %s: n %d  wall %.1f ms  rounds %d eps %.6g noise %d clusters %d  growth rounds %.1f ms  clustering %.1f ms (core %.2f union %.2f label %.2f)  build %.1f ms" % (
    os.path.basename(os.environ.get("OWL_MI355X_LIB", "default")), n, w, i["rounds"], i["eps"], i["noise"], i["clusters"], i["probe_ms"], i["solve_ms"],
    i["core_ms"], i["union_ms"], i["label_ms"], b["build_ms"]), flush=True)
