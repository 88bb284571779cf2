#!/usr/bin/env python3
"""Which clump generators make the round-3 library count a subtree twice?  Run with OWL_MI355X_LIB set to that
library on the GPU box; prints the number of wrong core flags per generator and minPts."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402  (a probe script: the checker is what it compares with)
from owlraytracing_amd.trueknn import TrueKNN  # noqa: E402


def clumps(rng, n_clumps, min_pts, eps, two_sided, lo, hi, depth, middle):
    out = []
    for c in rng.uniform(0.05, 0.95, (n_clumps, 3)):
        m = int(rng.integers(max(3, min_pts // 2), min_pts + 3))
        h = eps * rng.uniform(lo, hi)
        j = rng.integers(0, depth, (m, 1))
        sign = rng.choice([-1.0, 1.0], (m, 3)) if two_sided else rng.choice([-1.0, 1.0], 3)
        pts = c + sign * h * (0.5 ** j) * rng.uniform(0.7, 1.0, (m, 3))
        if middle:
            pts[0] = c
        out.append(pts)
    return np.concatenate(out).astype(np.float32)


def main():
    eps = float(np.float32(0.004))
    eng = TrueKNN()
    gens = {
        "one_sided": dict(two_sided=False, lo=0.30, hi=0.56, depth=9, middle=True),
        "two_sided": dict(two_sided=True, lo=0.30, hi=0.56, depth=9, middle=True),
        "two_sided_deep": dict(two_sided=True, lo=0.30, hi=0.56, depth=14, middle=True),
        "two_sided_wide": dict(two_sided=True, lo=0.25, hi=0.9, depth=9, middle=False),
    }
    for name, g in gens.items():
        for min_pts in (8, 10, 16, 33):
            rng = np.random.default_rng(7 * min_pts + len(name))
            xyz = clumps(rng, 20000, min_pts, eps, **g)
            ref = oracle.dbscan(xyz, eps, min_pts)
            eng.build(xyz)
            for paths in ("1", "0"):
                os.environ["TKNN_DB_PATHS"] = paths
                got = eng.dbscan(eps, min_pts)
                wrong = int((got["core"].cpu().numpy().astype(bool) != ref["core"].astype(bool)).sum())
                print("%-16s minPts=%2d paths=%s n=%d core=%d wrong=%d" % (name, min_pts, paths, len(xyz), int(ref["core"].sum()), wrong), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
