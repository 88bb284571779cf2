#!/usr/bin/env python3
"""Randomised parity sweep (GPU box): the three kernels against the replay checker on random sizes,
k, start radii and point distributions.  Test infrastructure: uses oracle/ as the checker.

    python scripts/fuzz_parity.py [seconds] [seed]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from conftest import assert_rows_equal  # noqa: E402
from owlraytracing_amd import _lib, datasets  # noqa: E402
from owlraytracing_amd.trueknn import TrueKNN  # noqa: E402


def make(rng, n):
    kind = rng.integers(0, 7)
    if kind == 0:
        return "uniform", datasets.uniform3d(n, seed=int(rng.integers(1 << 30)))
    if kind == 1:
        return "gmm", datasets.gaussian_mixture3d(n, components=int(rng.integers(1, 20)), sigma=float(10 ** rng.uniform(-3, -1)), seed=int(rng.integers(1 << 30)))
    if kind == 2:
        return "taxi2d", datasets.pad_to_3d(datasets.taxi_like2d(n, components=int(rng.integers(2, 40)), seed=int(rng.integers(1 << 30))))
    if kind == 3:
        x = datasets.uniform3d(n, seed=int(rng.integers(1 << 30)))
        m = max(1, n // int(rng.integers(2, 9)))
        x[rng.choice(n, m, replace=False)] = x[rng.integers(0, n, m)]
        return "duplicates", x
    if kind == 4:
        side = max(2, int(round(n ** (1 / 3))))
        g = np.arange(side, dtype=np.float32) * np.float32(0.0625)
        return "lattice", np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    if kind == 5:
        return "offset", (datasets.uniform3d(n, seed=int(rng.integers(1 << 30))) * np.float32(3) + np.float32(rng.choice([-700.0, 0.001, 55.5]))).astype(np.float32)
    return "band", datasets.boundary_band(max(2, n // 91), float(10 ** rng.uniform(-3, -1)), seed=int(rng.integers(1 << 30)))


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    t0 = time.time()
    cases = 0
    eng = TrueKNN()
    while time.time() - t0 < budget:
        n = int(10 ** rng.uniform(1.9, 5.2))
        # one case in seven above the register lists (k > 64: the team walk with the lists in memory), on sets the CPU replay
        # gets through in a second
        big_k = rng.random() < 0.15
        if big_k:
            n = min(n, 20_000)
        name, xyz = make(rng, n)
        n = len(xyz)
        k = int(rng.choice([65, 80, 100, 160, 300])) if big_k else int(rng.choice([1, 2, 3, 5, 8, 10, 16, 17, 24, 32, 33, 40, 47, 48, 49, 64]))
        kernels = (_lib.KERNEL_TEAM, _lib.KERNEL_WAVE, _lib.KERNEL_LANE) if k <= 64 else (_lib.KERNEL_TEAM,)  # (k > 64: the lists in memory, team walk only)
        if n <= k + 1:
            continue
        ext = float(np.ptp(xyz, axis=0).max()) or 1.0
        r0 = float(np.float32(0.25 * ext * (k / n) ** (1 / 3) * 10 ** rng.uniform(-1.3, 0.9)))
        try:
            ref = oracle.trueknn(xyz, k, r0, max_rounds=64)
        except Exception as e:  # e.g. duplicates-only sets that never reach k others
            print("skip %s n=%d k=%d r0=%g: %s" % (name, n, k, r0, e), flush=True)
            continue
        eng.build(xyz)
        for kern in kernels:
            r = eng.solve(k, r0, kernel=kern)
            tag = "%s n=%d k=%d r0=%g kernel=%d" % (name, n, k, r0, kern)
            assert r["info"]["rounds"] == ref["rounds"], tag
            assert np.array_equal(r["intersections"].cpu().numpy(), ref["intersections"]), tag
            try:
                assert_rows_equal(r["idx"].cpu().numpy(), r["dist"].cpu().numpy(), ref["idx"], ref["dist"])
            except AssertionError as e:
                raise AssertionError(tag + ": " + str(e))
        halo_note = ""
        if rng.random() < 0.35 and n > 4 * (k + 2):
            # the sharded layout: queries = the points on one side of a random plane (own tree, global
            # ids), the rest as the halo tree every query also searches
            axis, cut = int(rng.integers(0, 3)), float(np.quantile(xyz[:, int(rng.integers(0, 3))], rng.uniform(0.3, 0.7)))
            own = np.nonzero(xyz[:, axis] <= cut)[0].astype(np.int32)
            rest = np.nonzero(xyz[:, axis] > cut)[0].astype(np.int32)
            if len(own) > k + 1 and len(rest) > 0:
                sub = oracle.trueknn(xyz, k, r0, query_ids=own, max_rounds=64)
                eng.build(xyz[own], own)
                eng.set_halo(xyz[rest], rest)
                for kern in kernels:
                    r = eng.solve(k, r0, kernel=kern)
                    tag = "halo %s n=%d own=%d k=%d r0=%g kernel=%d" % (name, n, len(own), k, r0, kern)
                    assert np.array_equal(r["intersections"].cpu().numpy(), sub["intersections"][own]), tag
                    try:
                        assert_rows_equal(r["idx"].cpu().numpy(), r["dist"].cpu().numpy(), sub["idx"][own], sub["dist"][own])
                    except AssertionError as e:
                        raise AssertionError(tag + ": " + str(e))
                eng.set_halo(None, None)
                halo_note = " +halo(own=%d)" % len(own)
        cases += 1
        print("ok %-10s n=%6d k=%2d r0=%-10.4g rounds=%2d mean_isect=%.1f%s" % (name, n, k, r0, ref["rounds"], ref["intersections"].mean(), halo_note), flush=True)
    print("fuzz: %d cases x 3 kernels agree with the checker in %.0f s" % (cases, time.time() - t0))


if __name__ == "__main__":
    main()
