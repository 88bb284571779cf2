#!/usr/bin/env python3
"""Randomised parity sweep for RT-DBSCAN (GPU box): the HIP path against oracle/dbscan_oracle.c on random
sizes, eps, minPts and point distributions.  Test infrastructure: uses oracle/ as the checker.

    python scripts/fuzz_dbscan.py [seconds] [seed]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
import torch  # noqa: E402
from owlraytracing_amd import datasets  # noqa: E402
from owlraytracing_amd.trueknn import TrueKNN  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    t0 = time.time()
    cases = 0
    eng = TrueKNN()
    while time.time() - t0 < budget:
        n = int(10 ** rng.uniform(1.5, 5.3))
        kind = rng.integers(0, 6)
        forced_eps = None
        if kind == 0:
            name, xyz = "uniform", datasets.uniform3d(n, seed=int(rng.integers(1 << 30)))
        elif kind == 1:
            name, xyz = "gmm", datasets.gaussian_mixture3d(n, components=int(rng.integers(1, 30)), sigma=float(10 ** rng.uniform(-3, -1)), seed=int(rng.integers(1 << 30)))
        elif kind == 2:
            name, xyz = "taxi2d", datasets.pad_to_3d(datasets.taxi_like2d(n, components=int(rng.integers(2, 40)), seed=int(rng.integers(1 << 30))))
        elif kind == 3:
            name, xyz = "duplicates", datasets.uniform3d(n, seed=int(rng.integers(1 << 30)))
            m = max(1, n // int(rng.integers(2, 6)))
            xyz[rng.choice(n, m, replace=False)] = xyz[rng.integers(0, n, m)]
        elif kind == 4:
            # a lattice whose spacing IS eps (or a hair off): every axis neighbour sits within rounding of the threshold
            side = max(2, int(round(min(n, 60000) ** (1 / 3))))
            h = float(np.float32(10 ** rng.uniform(-2.5, -1)))
            g = np.arange(side, dtype=np.float32) * np.float32(h)
            xyz = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3).astype(np.float32) + np.float32(rng.uniform(0, 1))
            name, forced_eps = "lattice", float(np.float32(h * rng.choice([1.0, 1.0 - 1e-6, 1.0 + 1e-6, 1.42, 0.999])))
        else:
            # two dense slabs with a gap close to eps: groups face each other across it, most probes find nothing
            m = min(n, 100000) // 2 + 1
            gap = float(10 ** rng.uniform(-2.5, -1.3))
            a = rng.uniform(0, 1, (m, 3)).astype(np.float32) * np.float32([0.2, 0.2, 0.02])
            b = rng.uniform(0, 1, (m, 3)).astype(np.float32) * np.float32([0.2, 0.2, 0.02]) + np.float32([0, 0, 0.02 + gap])
            xyz = np.concatenate([a, b]).astype(np.float32)
            xyz = xyz[rng.permutation(len(xyz))]
            name, forced_eps = "slabs", float(np.float32(gap * rng.choice([0.98, 1.0, 1.02, 1.2])))
        min_pts = int(rng.choice([1, 2, 3, 4, 5, 8, 16, 40]))
        ext = float(np.ptp(xyz, axis=0).max()) or 1.0
        eps = float(np.float32(ext * (min_pts / len(xyz)) ** (1 / 3) * 10 ** rng.uniform(-0.7, 0.9)))
        if forced_eps is not None:
            eps = forced_eps
        ref = oracle.dbscan(xyz, eps, min_pts)
        eng.build(torch.from_numpy(xyz).cuda())
        got = eng.dbscan(eps, min_pts, want_counts=bool(rng.integers(0, 2)))
        tag = "%s n=%d eps=%g minPts=%d" % (name, len(xyz), eps, min_pts)
        assert np.array_equal(got["core"].cpu().numpy(), ref["core"].astype(bool)), tag
        assert np.array_equal(got["labels"].cpu().numpy(), ref["labels"]), tag
        assert got["info"]["clusters"] == int(ref["clusters"]), tag
        if "counts" in got:
            assert np.array_equal(got["counts"].cpu().numpy(), ref["counts"]), tag
        cases += 1
        print("ok %-10s n=%6d eps=%-10.4g minPts=%2d clusters=%d core=%d" % (name, len(xyz), eps, min_pts, int(ref["clusters"]), int(ref["core"].sum())), flush=True)
    print("fuzz: %d DBSCAN cases agree with the checker in %.0f s" % (cases, time.time() - t0))


if __name__ == "__main__":
    main()
