#!/usr/bin/env python3
"""GPU box: RT-DBSCAN at large coordinate magnitudes (the box prefilters' margins are absolute roundings of the coordinates,
the passes' bounds relative to eps): labels against the CPU spec on shifted copies of one set."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from owlraytracing_amd import datasets  # noqa: E402
from owlraytracing_amd.trueknn import TrueKNN  # noqa: E402

eng = TrueKNN()
bad = 0
for seed in range(6):
    base = datasets.gaussian_mixture3d(150_000, components=12, sigma=0.02, seed=seed)
    for off in (0.0, 55.5, -700.0, 4096.0, 30000.0):
        xyz = (base + np.float32(off)).astype(np.float32)
        for eps, mp in ((0.01, 4), (0.004, 6)):
            eps = float(np.float32(eps))
            ref = oracle.dbscan(xyz, eps, mp)
            eng.build(xyz)
            got = eng.dbscan(eps, mp)
            lab = got["labels"].cpu().numpy()
            wrong = int((lab != ref["labels"]).sum()) + int((got["core"].cpu().numpy().astype(bool) != ref["core"].astype(bool)).sum())
            bad += wrong > 0
            print("seed %d offset %8g eps %.4g minPts %d: clusters %d / %d, wrong %d" % (seed, off, eps, mp, got["info"]["clusters"], ref["clusters"], wrong), flush=True)
print("cases with a wrong label or core flag:", bad)
