#!/usr/bin/env python3
"""Developer timing loop: both kernels, several sizes.  Not the contract bench (see bench.py)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from owlraytracing_amd import _lib, datasets  # noqa: E402
from owlraytracing_amd.trueknn import TrueKNN  # noqa: E402


def main():
    sizes = [int(s) for s in (sys.argv[1] if len(sys.argv) > 1 else "1000000,10000000").split(",")]
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    kernels = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "1,2,3").split(",")]
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    for n in sizes:
        data = os.environ.get("QB_DATA", "uniform")
        if data == "gmm":  # BASELINE config 3's point set
            xyz, r0 = datasets.gaussian_mixture3d(n, components=64, sigma=0.02, seed=1), 0.0005
        elif data == "taxi":
            xyz, r0 = datasets.pad_to_3d(datasets.taxi_like2d(n, components=256, seed=2)), 0.0002
        else:
            xyz, r0 = datasets.uniform3d(n, seed=0), datasets.start_radius(n, k)
        r0 = float(os.environ.get("QB_R0", r0))
        xyz = torch.from_numpy(xyz).cuda()
        eng = TrueKNN()
        bi = eng.build(xyz)
        bi = eng.build(xyz)
        print("n=%d k=%d r0=%.5g build_ms=%.2f tree_MB=%.0f" % (n, k, r0, bi["build_ms"], bi["device_bytes"] / 1e6), flush=True)
        out = None
        for kern in kernels:
            best = None
            for _ in range(reps):
                torch.cuda.synchronize()
                t = time.perf_counter()
                r = eng.solve(k, r0, kernel=kern, out=out)
                torch.cuda.synchronize()
                wall = (time.perf_counter() - t) * 1e3
                out = {kk: v for kk, v in r.items() if kk != "info"}
                i = r["info"]
                if os.environ.get("QB_INFO"):
                    print("   ", i, flush=True)
                best = wall if best is None else min(best, wall)
            print("  kernel=%d wall_ms=%.2f dev_ms=%.2f main_kernel_ms=%.2f rounds=%d isect/q=%.1f node_tests=%.3g point_tests=%.3g q/s=%.3g" % (
                kern, best, i["solve_ms"], i["dominant_kernel_ms"], i["rounds"], i["total_intersections"] / n, i["node_tests"], i["point_tests"], n / best * 1e3), flush=True)
        eng.close()


if __name__ == "__main__":
    main()
