#!/usr/bin/env python3
"""GPU box: team-kernel solve time of BASELINE config 2's point set over k (TKNN_VERBOSE=1 adds the hand-over lines)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from owlraytracing_amd import datasets  # noqa: E402
from owlraytracing_amd.trueknn import TrueKNN  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ks = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "10,16,17,32,33,50,64").split(",")]
pts = torch.from_numpy(datasets.uniform3d(n, seed=0)).cuda()
eng = TrueKNN()
eng.build(pts)
for k in ks:
    r0 = datasets.start_radius(n, k)
    best = None
    for _ in range(2):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = eng.solve(k, r0, kernel=3)
        torch.cuda.synchronize()
        w = (time.perf_counter() - t) * 1e3
        best = w if best is None else min(best, w)
    i = r["info"]
    print("lib %s k=%d wall %.2f ms  device %.2f  main kernel %.2f  isect/q %.1f" % (
        os.path.basename(os.environ.get("OWL_MI355X_LIB", "default")), k, best, i["solve_ms"], i["dominant_kernel_ms"], i["total_intersections"] / n), flush=True)
    del r
