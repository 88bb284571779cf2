#!/usr/bin/env python3
"""Command line of the reference sample, served by the fused engine:

    tools/trueknn_cli.py <points.csv> <n> <dim> <start_radius|auto> <k> <timefile> [--out rows.npz]

Arguments as samples/s01-trueknn/hostCode.cpp:66-73; prints the same "Build time" / "True KNN time"
/ "Total time" lines (hostCode.cpp:211,344-347) and appends the total to <timefile> (:349-356).
`auto` takes the start radius from owlraytracing_amd.radius.sample_start_radius.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("file")
    ap.add_argument("n", type=int)
    ap.add_argument("dim", type=int)
    ap.add_argument("start_radius")
    ap.add_argument("k", type=int)
    ap.add_argument("timefile")
    ap.add_argument("--out", default=None, help="write idx/dist/intersections/levels as .npz (tools/compare_reference_dump.py --rows reads it)")
    ap.add_argument("--kernel", type=int, default=0)
    a = ap.parse_args()

    from owlraytracing_amd import datasets
    from owlraytracing_amd.radius import sample_start_radius
    from owlraytracing_amd.trueknn import TrueKNN

    pts = datasets.pad_to_3d(datasets.read_csv_points(a.file, a.n, a.dim))
    print("#owl.sample(main):  num spheres: %d" % len(pts))
    r0 = sample_start_radius(pts) if a.start_radius == "auto" else float(a.start_radius)
    eng = TrueKNN()
    b = eng.build(pts)
    print("Build time: %g seconds." % (b["build_ms"] / 1e3))
    r = eng.solve(a.k, r0, kernel=a.kernel, want_levels=True)
    info = r["info"]
    print("Rounds: %d  start radius %g  final radius %g" % (info["rounds"], r0, info["final_radius"]))
    print("True KNN time: %g seconds." % (info["solve_ms"] / 1e3))
    tot = (b["build_ms"] + info["solve_ms"]) / 1e3
    print("Total time: %g" % tot)
    with open(a.timefile, "a") as fh:
        fh.write("%g\n" % tot)
    if a.out:
        np.savez(a.out, idx=r["idx"].cpu().numpy(), dist=r["dist"].cpu().numpy(),
                 intersections=r["intersections"].cpu().numpy(), levels=r["levels"].cpu().numpy())
    eng.close()


if __name__ == "__main__":
    main()
