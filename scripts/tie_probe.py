"""GPU check: rows on tie-heavy sets equal the call-by-call replay exactly (index for index)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import oracle
from owlraytracing_amd import _lib, datasets
from owlraytracing_amd.trueknn import TrueKNN

def lattice(m, dims=3, seed=0, drop=0.2):
    g = np.arange(m, dtype=np.float32) / np.float32(32)
    if dims == 3:
        xyz = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    else:
        xy = np.stack(np.meshgrid(g, g, indexing="ij"), -1).reshape(-1, 2)
        xyz = np.concatenate([xy, np.zeros((len(xy), 1), np.float32)], 1)
    rng = np.random.default_rng(seed)
    xyz = xyz[rng.random(len(xyz)) > drop]
    return np.ascontiguousarray(xyz[rng.permutation(len(xyz))])

cases = [("crossround", datasets.cross_round_ties(), 2, 1.0)]
for k in (1, 3, 5, 8, 15, 16, 17, 31, 32, 33, 48, 64):
    cases.append(("lattice3d", lattice(14, 3, seed=k), k, 0.02))
    cases.append(("lattice2d", lattice(50, 2, seed=k), k, 0.011))
x = datasets.gaussian_mixture3d(30_000, components=8, sigma=0.01, seed=5)
x[::7] = x[3::7][: len(x[::7])]
for k in (5, 16, 32):
    cases.append(("dups", x, k, 0.002))
q = np.round(datasets.uniform3d(40_000, seed=3) * 64) / 64  # heavy quantisation: duplicates and lattice ties
for k in (5, 16, 24):
    cases.append(("quantised", q.astype(np.float32), k, 0.01))
bad = 0
for name, xyz, k, r0 in cases:
    ref = oracle.trueknn(xyz, k, r0)
    for kern, kname in ((_lib.KERNEL_TEAM, "team"), (_lib.KERNEL_WAVE, "wave"), (_lib.KERNEL_LANE, "lane")):
        for tail in (None, "walk", "lane") if kern == _lib.KERNEL_TEAM else (None,):
            if tail: os.environ["TKNN_TEAM_TAIL"] = tail
            else: os.environ.pop("TKNN_TEAM_TAIL", None)
            eng = TrueKNN(); eng.build(xyz)
            r = eng.solve(k, r0, kernel=kern)
            idx, dist = r["idx"].cpu().numpy(), r["dist"].cpu().numpy()
            ok_d = np.array_equal(dist.view(np.int32), ref["dist"].view(np.int32))
            wrong = int((idx != ref["idx"]).any(axis=1).sum())
            ok_i = np.array_equal(r["intersections"].cpu().numpy(), ref["intersections"])
            info = r["info"]
            status = "ok" if (ok_d and wrong == 0 and ok_i) else "MISMATCH"
            bad += status != "ok"
            print(f"{name:10s} n={len(xyz):6d} k={k:2d} {kname}{'/'+tail if tail else '':5s} rounds={info['rounds']} ties={info['tie_rows']} left={info['tie_rows_left']} "
                  f"tie_ms={info['tie_ms']:.3f} dist_ok={ok_d} rows_wrong={wrong} isect_ok={ok_i} {status}", flush=True)
            eng.close()
print("MISMATCHES:", bad)
sys.exit(1 if bad else 0)
