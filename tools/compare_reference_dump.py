#!/usr/bin/env python3
"""Compare a neighbour dump of the reference (an OptiX run of samples/s01-trueknn) with this engine.

The reference can print its result rows: `/root/reference/samples/s01-trueknn/hostCode.cpp:312-321`
holds, commented out, a loop writing one line per neighbour slot

    j,ind,dist            (outfile << j << "," << fb[j*knn+i].ind << ',' << fb[j*knn+i].dist << '\\n')

optionally preceded by a "Point j: (x, y, z)" line per query.  Nothing in this repository can produce
that file (nvcc + the closed OptiX SDK are needed); somebody with an NVIDIA box can -- INTEGRATION.md,
section "Pinning parity against an OptiX run", says how -- and this tool then gives the verdict the
parity claim of DESIGN.md section 1 is waiting for:

    tools/compare_reference_dump.py points.csv <n> <dim> <start_radius> <k> dump.txt [--rows rows.npz]

Without --rows the engine is run here (needs an MI355X); with it, rows solved earlier are compared
(`tools/trueknn_cli.py ... --out rows.npz`; the arrays idx, dist and, if present, levels).

The dump loop sits inside the round loop and stops at the first unfinished query, so a dump holds the
partial lists of every round: the LAST k lines of a query are its final row.  `operator<<(float)` prints
six significant digits unless the reference user raises the precision; distances are compared at the
precision the file has (exactly, in fp32 ulps, when it has nine digits).

Every row gets one verdict:

  identical     same indices in the same order, distances equal (within --ulps at full precision, within
                the printed precision otherwise)
  tie-order     same index multiset; positions differ only inside runs of equal distances (the reference's
                order of exact ties depends on RT-core visit order, deviceCode.cu:116,125 strict '<')
  distance      same indices and order, a distance differs by more than --ulps but by less than
                --max-ulps (the reference's Release build is --use_fast_math: contracted fmas and
                sqrt.approx.ftz, owl/cmake/configure_optix.cmake:49-53)
  box-face      the index sets differ, but the reference's row is the k nearest of the query's candidates
                at its final level (or one level before / after) for SOME decision about the candidates
                lying within the band |t - r_l| <= band of a face of that box (t = Chebyshev distance to
                the query, r_l = start_radius * 2^l): the points on which a conservative hardware ray/box
                test and the closed fp32 box of DESIGN.md section 1 may disagree -- including rows that
                finished one level apart because such a candidate decided whether a box held k others
  mismatch      anything else

Exit status 0 iff no row is a mismatch.  A JSON summary goes to stdout (and to --json).
"""
import argparse
import json
import os
import re
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

_LINE = re.compile(r"^\s*(-?\d+)\s*,\s*(-?\d+)\s*,\s*([-+0-9.eEinfINFnaNA]+)\s*$")


def parse_dump(path_or_lines, n, k):
    """-> (idx[n,k] int64, dist[n,k] float64, text[n,k] object, seen[n] bool): the last k lines of every query."""
    idx = np.full((n, k), -2, np.int64)
    dist = np.full((n, k), np.nan, np.float64)
    text = np.empty((n, k), object)
    fill = np.zeros(n, np.int64)  # lines seen for the query in the current group
    last_q = -1
    lines = open(path_or_lines) if isinstance(path_or_lines, str) else path_or_lines
    for line in lines:
        m = _LINE.match(line)
        if not m:
            continue  # "Point j: (...)" and anything else the user's edit prints
        q = int(m.group(1))
        if q < 0 or q >= n:
            raise ValueError("dump names query %d, but n = %d" % (q, n))
        if q != last_q:
            fill[q] = 0  # a new group of this query (a later round): overwrite
            last_q = q
        slot = int(fill[q] % k)
        if fill[q] and slot == 0:
            pass  # more than k lines in one run: keep overwriting round-robin (the last k survive)
        idx[q, slot] = int(m.group(2))
        token = m.group(3)
        dist[q, slot] = float(token)
        text[q, slot] = token
        fill[q] += 1
    if hasattr(lines, "close"):
        lines.close()
    seen = (idx != -2).all(axis=1)
    return idx, dist, text, seen


def significant_digits(tokens):
    """Largest number of significant decimal digits among the printed distances (6 = iostream default)."""
    best = 0
    for t in tokens:
        if t is None:
            continue
        mant = re.split(r"[eE]", t)[0].lstrip("+-")
        if any(c.isalpha() for c in mant):
            continue
        digits = mant.replace(".", "").lstrip("0")
        best = max(best, len(digits))
    return best


def ulp_distance(a, b):
    """fp32 ulps between two arrays of non-negative floats (ordered-integer image)."""
    ia = np.asarray(a, np.float32).view(np.int32).astype(np.int64)
    ib = np.asarray(b, np.float32).view(np.int32).astype(np.int64)
    return np.abs(ia - ib)


def explained_by_face_candidates(points, q, k, r0, level, ref_row, band_ulps, max_subsets=4096):
    """Is `ref_row` (an index set) the k nearest of the query's candidate set at level-1, level or
    level+1 for SOME decision about the candidates lying on a face of that box?

    certain(l) = points with Chebyshev distance t <= r_l - band, face(l) = those with |t - r_l| <= band
    (band = band_ulps * 2^-23 (max|q| + 2 r_l)).  A run may end at level l if it cannot have ended
    earlier for certain (|certain(l-1)| < k others) and certain(l) plus some subset S of face(l) holds at
    least k others; its row is then the k nearest of certain(l) + S (engine distance arithmetic,
    deviceCode.cu:110-113).  Order inside the row is not compared here (the caller handles ties)."""
    import itertools

    pts = points.astype(np.float32)
    qp = pts[q]
    d = pts - qp
    t_all = np.abs(d).max(axis=1)
    dist = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2], dtype=np.float32)
    mag = float(np.abs(qp).max())
    want = set(int(v) for v in ref_row if v >= 0)
    if len(want) != k:
        return False
    others = np.ones(len(pts), bool)
    others[q] = False
    for l in (level - 1, level, level + 1):
        if l < 0:
            continue
        r = float(np.float32(r0)) * 2.0 ** l
        band = band_ulps * 2.0 ** -23 * (mag + 2.0 * r)
        if l > 0:
            r_prev = r / 2.0
            band_prev = band_ulps * 2.0 ** -23 * (mag + 2.0 * r_prev)
            if int(((t_all <= r_prev - band_prev) & others).sum()) >= k:
                continue  # certainly finished before level l
        certain = np.nonzero((t_all <= r - band) & others)[0]
        face = np.nonzero((np.abs(t_all - r) <= band) & others)[0]
        if len(face) == 0 or len(certain) + len(face) < k:
            continue
        tried = 0
        for size in range(len(face) + 1):
            for sub in itertools.combinations(face.tolist(), size):
                tried += 1
                if tried > max_subsets:
                    break
                cand = np.concatenate([certain, np.asarray(sub, np.int64)]) if sub else certain
                if len(cand) < k:
                    continue
                dd = dist[cand]
                kth = np.partition(dd, k - 1)[k - 1]
                sure = set(cand[dd < kth].tolist())      # strictly nearer than the k-th: must be in the row
                tied = set(cand[dd == kth].tolist())     # at the k-th distance: any of them may fill it up
                if sure <= want and want <= (sure | tied):
                    return True
            if tried > max_subsets:
                break
    return False


def classify(points, k, r0, ref_idx, ref_dist, digits, eng_idx, eng_dist, eng_levels=None, ulps=0, max_ulps=64,
             band_ulps=8.0, rows=None):
    """Row verdicts (see the module docstring).  Returns dict(verdict=array of str, detail=dict)."""
    n = len(points)
    rows = np.arange(n) if rows is None else np.asarray(rows)
    verdict = np.empty(len(rows), object)
    full_precision = digits >= 9
    rel = 0.5 * 10.0 ** (1 - max(digits, 1)) if not full_precision else 0.0
    worst_ulps = 0
    r0 = np.float32(r0)

    def dist_equal(a, b, tol_ulps):
        if full_precision:
            return ulp_distance(a, b) <= tol_ulps
        a = np.asarray(a, np.float64)
        b = np.asarray(b, np.float64)
        # the printed value is the engine's value rounded to `digits` digits, up to tol_ulps of fp32 slack
        slack = rel * np.maximum(np.abs(a), np.abs(b)) + tol_ulps * np.spacing(np.asarray(b, np.float32)).astype(np.float64)
        return np.abs(a - b) <= slack + 1e-45

    for out, q in enumerate(rows):
        ri, rd = ref_idx[q], ref_dist[q]
        ei, ed = np.asarray(eng_idx[q], np.int64), np.asarray(eng_dist[q], np.float32)
        same_order = np.array_equal(ri, ei)
        if same_order:
            if dist_equal(rd, ed, ulps).all():
                verdict[out] = "identical"
                continue
            if dist_equal(rd, ed, max_ulps).all():
                verdict[out] = "distance"
                if full_precision:
                    worst_ulps = max(worst_ulps, int(ulp_distance(rd, ed).max()))
                continue
            verdict[out] = "mismatch"
            continue
        if sorted(ri.tolist()) == sorted(ei.tolist()):
            # same neighbours: do the positions differ only inside runs of (engine) equal distances?
            ok = True
            for pos in np.nonzero(ri != ei)[0]:
                run = np.nonzero(ed.view(np.int32) == ed.view(np.int32)[pos])[0]
                if len(run) < 2 or sorted(ri[run].tolist()) != sorted(ei[run].tolist()):
                    # with a fast-math sqrt two nearly equal distances may also swap: accept within max_ulps
                    near = np.nonzero(ulp_distance(ed, np.full(k, ed[pos], np.float32)) <= max_ulps)[0]
                    if len(near) < 2 or sorted(ri[near].tolist()) != sorted(ei[near].tolist()):
                        ok = False
                        break
            verdict[out] = "tie-order" if ok and dist_equal(np.sort(rd), np.sort(ed.astype(np.float64)), max_ulps).all() else "mismatch"
            continue
        # different neighbour sets: only box-face candidates may explain that
        level = None if eng_levels is None else int(eng_levels[q])
        if level is None or level < 0:
            verdict[out] = "mismatch"
            continue
        verdict[out] = "box-face" if explained_by_face_candidates(points, q, k, r0, level, ri, band_ulps) else "mismatch"
    counts = {name: int((verdict == name).sum()) for name in ("identical", "tie-order", "distance", "box-face", "mismatch")}
    return {"verdict": verdict, "counts": counts, "worst_distance_ulps": worst_ulps,
            "distance_precision_digits": digits}


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("points", help="the CSV the reference was run on (hostCode.cpp:83-104 format)")
    ap.add_argument("n", type=int)
    ap.add_argument("dim", type=int)
    ap.add_argument("start_radius", type=float)
    ap.add_argument("k", type=int)
    ap.add_argument("dump", help="text the reference wrote with hostCode.cpp:312-321 uncommented")
    ap.add_argument("--rows", default=None, help=".npz with idx, dist[, levels] solved earlier (no GPU needed)")
    ap.add_argument("--ulps", type=int, default=0, help="distance slack for 'identical' (fp32 ulps)")
    ap.add_argument("--max-ulps", type=int, default=64, help="distance slack for 'distance' / near-tie swaps (fast-math)")
    ap.add_argument("--band-ulps", type=float, default=8.0,
                    help="half-width of the box-face band in units of 2^-23 (|q| + 2r); 4 = the engine's own 2M band")
    ap.add_argument("--json", default=None)
    ap.add_argument("--show", type=int, default=10, help="print that many non-identical rows")
    a = ap.parse_args(argv)

    from owlraytracing_amd import datasets

    pts = datasets.pad_to_3d(datasets.read_csv_points(a.points, a.n, a.dim))
    n = len(pts)
    ref_idx, ref_dist, text, seen = parse_dump(a.dump, n, a.k)
    digits = significant_digits(text[seen].ravel().tolist())
    if a.rows:
        z = np.load(a.rows)
        eng_idx, eng_dist = z["idx"], z["dist"]
        levels = z["levels"] if "levels" in z.files else None
    else:
        from owlraytracing_amd.trueknn import TrueKNN

        eng = TrueKNN()
        eng.build(pts)
        r = eng.solve(a.k, a.start_radius, want_levels=True)
        eng_idx, eng_dist, levels = r["idx"].cpu().numpy(), r["dist"].cpu().numpy(), r["levels"].cpu().numpy()
        eng.close()
    rows = np.nonzero(seen)[0]
    res = classify(pts, a.k, a.start_radius, ref_idx, ref_dist, digits, eng_idx, eng_dist, levels, a.ulps, a.max_ulps,
                   a.band_ulps, rows=rows)
    summary = {"rows_in_dump": int(seen.sum()), "rows_missing_from_dump": int(n - seen.sum()), "k": a.k,
               "start_radius": a.start_radius, **{kk: v for kk, v in res.items() if kk != "verdict"}}
    summary["parity"] = "pinned" if res["counts"]["mismatch"] == 0 and seen.all() else "NOT pinned"
    shown = 0
    for q, v in zip(rows, res["verdict"]):
        if v != "identical" and shown < a.show:
            print("row %d: %s\n   reference %s %s\n   engine    %s %s" % (q, v, ref_idx[q].tolist(), ref_dist[q].tolist(),
                                                                       np.asarray(eng_idx[q]).tolist(), np.asarray(eng_dist[q]).tolist()),
                  file=sys.stderr)
            shown += 1
    print(json.dumps(summary))
    if a.json:
        with open(a.json, "w") as fh:
            json.dump(summary, fh)
    return 0 if res["counts"]["mismatch"] == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
