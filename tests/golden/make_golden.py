#!/usr/bin/env python3
"""Generate tests/golden/*.npz -- small input/output vectors for the TrueKNN path.

The reference has no fixtures for this path and cannot run here (PARITY UNPINNED, see
oracle/trueknn_oracle.c), so these vectors come from the C restatement and are accepted only when
the independent numpy restatement (oracle/trueknn_numpy.py) agrees bit for bit on every index,
distance and intersection count and on the round count.  Each file holds
  xyz (n,3) f32 | k | start_radius | idx (n,k) i32 | dist (n,k) f32 | intersections (n,) i64 |
  rounds | final_radius | order_free (n,) bool
``order_free`` marks rows that come out identical under ascending, descending and shuffled
candidate visit orders, i.e. rows a traversal-order-dependent implementation (like the reference
itself, deviceCode.cu:116,125 strict '<') must reproduce exactly.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from oracle.trueknn_numpy import trueknn_numpy  # noqa: E402
from owlraytracing_amd import datasets  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def cases():
    rng = np.random.default_rng(1234)
    yield "uniform3d_n2048_k5", datasets.uniform3d(2048, seed=0), 5, datasets.start_radius(2048, 5)
    yield "uniform3d_n4096_k10", datasets.uniform3d(4096, seed=3), 10, datasets.start_radius(4096, 10)
    yield "gmm3d_n3000_k10", datasets.gaussian_mixture3d(3000, components=8, sigma=0.03, seed=1), 10, 0.004
    yield "planar2d_n2048_k5", datasets.pad_to_3d(rng.random((2048, 2), dtype=np.float32)), 5, 0.006
    dup = datasets.uniform3d(1024, seed=5)
    dup[rng.choice(1024, 256, replace=False)] = dup[rng.integers(0, 1024, 256)]
    yield "duplicates_n1024_k4", dup, 4, 0.02
    g = np.arange(12, dtype=np.float32) * np.float32(0.125)
    lattice = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    yield "lattice_n1728_k6", lattice, 6, 0.125  # every box boundary is hit exactly; all ties
    yield "tinyradius_n1000_k3", datasets.uniform3d(1000, seed=7), 3, 1e-5
    yield "hugeradius_n1500_k8", datasets.uniform3d(1500, seed=8), 8, 2.0
    yield "k1_n777", datasets.uniform3d(777, seed=9), 1, 0.01
    yield "minimal_n6_k5", datasets.uniform3d(6, seed=10), 5, 0.05
    yield "taxi2d_n3000_k7", datasets.pad_to_3d(datasets.taxi_like2d(3000, components=16, seed=2)), 7, 0.001
    yield "k32_n2500", datasets.uniform3d(2500, seed=11), 32, 0.03
    # k above the engine's register lists (the reference takes any k, hostCode.cpp:111): the team walk with the lists in memory
    yield "k65_n1000", datasets.uniform3d(1000, seed=12), 65, 0.05
    yield "k100_n800", datasets.gaussian_mixture3d(800, components=4, sigma=0.05, seed=13), 100, 0.02
    yield "k256_n600", datasets.uniform3d(600, seed=14), 256, 0.1
    # satellites on the box faces of 40 anchors, +-2 ulps, coordinates from 1e-3 to 600
    yield "boundaryband_n3640_k7", datasets.boundary_band(40, 0.01, seed=4), 7, 0.01
    # exact-distance ties between candidates first seen in different rounds (persistent lists)
    yield "crossroundties_n400_k2", datasets.cross_round_ties(100, seed=17), 2, 1.0


def main():
    for name, xyz, k, r0 in cases():
        ref = oracle.trueknn(xyz, k, r0, order=oracle.ORDER_ASCENDING)
        chk = trueknn_numpy(xyz, k, r0)
        assert ref["rounds"] == chk["rounds"], name
        assert np.array_equal(ref["intersections"], chk["intersections"]), name
        assert np.array_equal(ref["idx"], chk["idx"]), name
        assert np.array_equal(ref["dist"], chk["dist"]), name
        assert np.all(ref["num_neighbors"] == 0), name
        order_free = np.ones(len(xyz), bool)
        variants = [oracle.trueknn(xyz, k, r0, order=oracle.ORDER_DESCENDING)]
        variants += [oracle.trueknn(xyz, k, r0, order=oracle.ORDER_SHUFFLED, seed=s) for s in (1, 2, 3)]
        for v in variants:
            assert np.array_equal(v["intersections"], ref["intersections"]), name
            assert v["rounds"] == ref["rounds"], name
            # whatever the order, the multiset of distances of a row is the same
            assert np.array_equal(v["dist"], ref["dist"]), name
            order_free &= np.all(v["idx"] == ref["idx"], axis=1)
        np.savez_compressed(
            os.path.join(OUT, name + ".npz"), xyz=xyz, k=np.int32(k), start_radius=np.float32(r0),
            idx=ref["idx"], dist=ref["dist"], intersections=ref["intersections"],
            rounds=np.int32(ref["rounds"]), final_radius=np.float32(ref["final_radius"]),
            order_free=order_free)
        print("%-24s n=%5d k=%2d r0=%.6g rounds=%d mean_isect=%.1f order_free=%d/%d" % (
            name, len(xyz), k, r0, ref["rounds"], ref["intersections"].mean(),
            order_free.sum(), len(xyz)))


if __name__ == "__main__":
    main()
