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
which = sys.argv[3] if len(sys.argv) > 3 else "uniform"
if which == "gmm":
    host, r_fixed = datasets.gaussian_mixture3d(n, components=64, sigma=0.02, seed=1), 0.0005
elif which == "taxi":
    host, r_fixed = datasets.pad_to_3d(datasets.taxi_like2d(n, components=256, seed=2)), 0.0002
else:
    host, r_fixed = datasets.uniform3d(n, seed=0), None
pts = torch.from_numpy(host).cuda()
eng = TrueKNN()
eng.build(pts)
for k in ks:
    r0 = r_fixed or datasets.start_radius(n, k)
    best = None
    r = None
    for _ in range(2):
        r = None
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = eng.solve(k, r0, kernel=3)
        torch.cuda.synchronize()
        w = (time.perf_counter() - t) * 1e3
        best = w if best is None else min(best, w)
    i = r["info"]
    print("lib %s %s k=%d wall %.2f ms  device %.2f  main kernel %.2f  isect/q %.1f" % (
        os.path.basename(os.environ.get("OWL_MI355X_LIB", "default")), which, k, best, i["solve_ms"], i["dominant_kernel_ms"], i["total_intersections"] / n), flush=True)
    del r
