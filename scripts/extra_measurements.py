#!/usr/bin/env python3
"""Side measurements for DESIGN.md (not the contract bench): non-uniform TrueKNN inputs, RT-DBSCAN at
BASELINE config 3 scale, and the unchanged reference sample through the OWL program model.  The two runs README / DESIGN
quote a time for -- BASELINE config 4's 100 M-point set and config 5's 50 M-point set on one GPU -- assert the structure of
what they time (check_rows; noise / cluster bookkeeping); their comparison with the CPU checker is in the GPU test suite."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from owlraytracing_amd import datasets  # noqa: E402
from owlraytracing_amd.trueknn import TrueKNN  # noqa: E402

out = {}


def check_rows(pts_dev, r, n, k):
    """What a timed solve must satisfy before its time is recorded (no checker here -- scripts never import oracle/; the
    comparisons with the CPU checker at these sizes are tests/test_trueknn_gpu.py::test_full_size_*): every row without its
    own point, indices in range, distances ascending, intersection counts adding up, and the distances of a 2 M-row slice
    recomputed in fp64 from the indices."""
    idx, dist, isect, info = r["idx"], r["dist"], r["intersections"], r["info"]
    assert info["unfinished"] == 0 and info["tie_rows_left"] == 0
    assert int(isect.sum()) == info["total_intersections"]
    step = 10_000_000
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        i, d = idx[lo:hi], dist[lo:hi]
        ar = torch.arange(lo, hi, device=i.device, dtype=torch.int32)[:, None]
        assert bool((i != ar).all()) and bool((i >= 0).all()) and bool((i < n).all())
        assert bool((d[:, 1:] >= d[:, :-1]).all())
    lo, hi = max(0, n // 2 - 1_000_000), min(n, n // 2 + 1_000_000)
    d64 = (pts_dev[idx[lo:hi].long()].double() - pts_dev[lo:hi, None, :].double()).norm(dim=2)
    assert float((d64 - dist[lo:hi].double()).abs().max()) < 1e-6


def timed_solve(name, pts, k, r0, kernels=(3, 2, 1), check=False):
    eng = TrueKNN()
    pts_dev = torch.from_numpy(pts).cuda()
    b = eng.build(pts_dev)
    res = {"n": len(pts), "k": k, "r0": r0, "build_ms": b["build_ms"]}
    for kern in kernels:
        best = None
        r = None
        for _ in range(2):
            r = None  # (its rows go back to torch's cache before the next call asks for the same sizes: no hipMalloc in the timed call)
            torch.cuda.synchronize()
            t = time.perf_counter()
            r = eng.solve(k, r0, kernel=kern)
            torch.cuda.synchronize()
            w = (time.perf_counter() - t) * 1e3
            best = w if best is None else min(best, w)
        if check:
            check_rows(pts_dev, r, len(pts), k)
        res["kernel_%d" % kern] = {"wall_ms": best, "device_ms": r["info"]["solve_ms"], "rounds": r["info"]["rounds"], "used": r["info"]["kernel_used"],
                                   "isect_per_query": r["info"]["total_intersections"] / len(pts), "rows_checked": bool(check)}
        print(name, kern, res["kernel_%d" % kern], flush=True)
    eng.close()
    out[name] = res


def main():
    which = sys.argv[1:] or ["gmm", "taxi", "ksweep", "sizes", "dbscan", "dbscan_auto", "sample"]
    if "ksweep" in which:  # BASELINE config 2's set at other k (team kernels: 1, 2, 3 or 4 list registers per lane)
        pts = datasets.uniform3d(10_000_000, seed=0)
        for k in (5, 16, 17, 32, 33, 40, 48, 50, 64, 65, 100):  # (k > 64: the team walk with the lists in memory)
            timed_solve("trueknn_uniform3d_10M_k%d" % k, pts, k, datasets.start_radius(len(pts), k), kernels=(3,))
        del pts
        pts = datasets.uniform3d(1_000_000, seed=0)
        for k in (256, 1024):
            timed_solve("trueknn_uniform3d_1M_k%d" % k, pts, k, datasets.start_radius(len(pts), k), kernels=(3,))
        del pts
    if "sizes" in which:  # config 4's whole 100 M-point set on ONE GPU
        for n in (50_000_000, 100_000_000):
            pts = datasets.uniform3d_counter(0, n, seed=0)
            timed_solve("trueknn_uniform3d_%dM_k10" % (n // 1_000_000), pts, 10, datasets.start_radius(n, 10), kernels=(3,), check=True)
            del pts
    if "dbscan_auto" in which:  # config 5's point set (50 M heavy-tailed 2-D points, 5 % duplicates) on ONE GPU, eps auto-grown
        pts = datasets.pad_to_3d(datasets.taxi_like2d(50_000_000, components=256, seed=2))
        eng = TrueKNN()
        b = eng.build(torch.from_numpy(pts).cuda())
        eng.dbscan_auto(0.00005, 4, 0.05)  # (untimed: the first call allocates the engine's workspace and the side stream)
        for eps0, max_noise in ((0.00005, 0.05), (0.00005, 0.01)):
            torch.cuda.synchronize()
            t = time.perf_counter()
            r = eng.dbscan_auto(eps0, 4, max_noise)
            torch.cuda.synchronize()
            w = (time.perf_counter() - t) * 1e3
            i = r["info"]
            # what the timed call must satisfy (the comparison with the CPU spec: tests/test_dbscan.py::test_config5_set_full_size_auto_eps)
            lab, core = r["labels"], r["core"]
            assert i["noise"] == int((lab < 0).sum()) <= int(max_noise * len(pts))
            assert i["clusters"] == int(lab.max()) + 1 and bool((lab[core] >= 0).all())
            out["dbscan_auto_taxi2d_50M_minpts4_eps0_%g_noise_%g" % (eps0, max_noise)] = {
                "wall_ms": w, "build_ms": b["build_ms"], "rounds": i["rounds"], "eps": i["eps"], "noise": i["noise"], "clusters": i["clusters"],
                "probe_rounds_ms": i["probe_ms"], "final_clustering_ms": i["solve_ms"]}
            print("dbscan_auto", out["dbscan_auto_taxi2d_50M_minpts4_eps0_%g_noise_%g" % (eps0, max_noise)], flush=True)
        eng.close()
        del pts
    if "gmm" in which:
        pts = datasets.gaussian_mixture3d(10_000_000, components=64, sigma=0.02, seed=1)
        timed_solve("trueknn_gmm3d_10M_k10", pts, 10, 0.0005)
    if "taxi" in which:
        pts = datasets.pad_to_3d(datasets.taxi_like2d(10_000_000, components=256, seed=2))
        timed_solve("trueknn_taxi2d_10M_k10", pts, 10, 0.0002)
    if "dbscan" in which:
        for n in (1_000_000, 10_000_000):
            pts = datasets.gaussian_mixture3d(n, components=64, sigma=0.02, seed=1)
            eng = TrueKNN()
            eng.build(torch.from_numpy(pts).cuda())
            torch.cuda.synchronize()
            t = time.perf_counter()
            r = eng.dbscan(float(np.float32(0.01)), 4)
            torch.cuda.synchronize()
            w = (time.perf_counter() - t) * 1e3
            lab = r["labels"]
            out["dbscan_gmm3d_%d_eps0.01_minpts4" % n] = {
                "wall_ms": w, "device_ms": r["info"]["solve_ms"], "clusters": r["info"]["clusters"],
                "noise": int((lab < 0).sum()), "core": int(r["core"].sum()),
                "kernels_ms": {"core_flags": r["info"]["core_ms"], "unions": r["info"]["union_ms"], "labels": r["info"]["label_ms"]},
                "node_tests": r["info"]["node_tests"], "point_tests": r["info"]["point_tests"]}
            print("dbscan", n, out["dbscan_gmm3d_%d_eps0.01_minpts4" % n], flush=True)
            eng.close()
    if "sample" in which:
        exe = os.path.join(ROOT, "oracle", "_ref", "sample01-trueknn")
        if os.path.exists(exe):
            for n in (1_000_000,):
                pts = datasets.uniform3d(n, seed=0)
                csv = "/tmp/pts_%d.csv" % n
                np.savetxt(csv, pts, fmt="%.9g", delimiter=",")
                r0 = datasets.start_radius(n, 10)
                p = subprocess.run([exe, csv, str(n), "3", repr(r0), "10", "/tmp/time.txt"], capture_output=True, text=True, timeout=900)
                lines = [l for l in p.stdout.splitlines() if "time" in l.lower() or l.startswith("Round:")]
                out["unchanged_reference_sample_n%d_k10" % n] = {"rc": p.returncode, "lines": lines[-8:]}
                print("sample", n, lines[-8:], flush=True)
    path = os.path.join(ROOT, "gpurun_out", "extra_measurements.json")
    if len(sys.argv) > 1 and os.path.exists(path):  # a partial run refreshes its entries of the file
        merged = json.load(open(path))
        merged.update(out)
        out.clear()
        out.update(merged)
    json.dump(out, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
