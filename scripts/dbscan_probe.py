#!/usr/bin/env python3
"""Developer probe: RT-DBSCAN at BASELINE config 3 (run under rocprofv3 --kernel-trace --stats for per-kernel times)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from owlraytracing_amd import datasets  # noqa: E402
from owlraytracing_amd.trueknn import TrueKNN  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
pts = datasets.gaussian_mixture3d(n, components=64, sigma=0.02, seed=1)
eng = TrueKNN()
eng.build(torch.from_numpy(pts).cuda())
for _ in range(2):
    torch.cuda.synchronize()
    t = time.perf_counter()
    r = eng.dbscan(float(np.float32(0.01)), 4)
    torch.cuda.synchronize()
    print("dbscan n=%d: %.1f ms wall, %.1f ms device, %d clusters" % (n, (time.perf_counter() - t) * 1e3, r["info"]["solve_ms"], r["info"]["clusters"]), flush=True)
