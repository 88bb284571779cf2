"""ctypes binding of oracle/libtrueknn_oracle.so (built by ``make -C oracle``)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libtrueknn_oracle.so")

ORDER_ASCENDING, ORDER_DESCENDING, ORDER_SHUFFLED = 0, 1, 2

# samples/s01-trueknn/GeomTypes.h:22-28 -- 24-byte record, 4 bytes of padding before the int64
NEIGH_DTYPE = np.dtype(
    {
        "names": ["ind", "dist", "numNeighbors", "intersections"],
        "formats": [np.int32, np.float32, np.int32, np.int64],
        "offsets": [0, 4, 8, 16],
        "itemsize": 24,
    }
)


class OracleError(RuntimeError):
    pass


def build(force=False):
    """Compile the C restatement (gcc); building the checker is not using it."""
    src = os.path.join(_HERE, "trueknn_oracle.c")
    src2 = os.path.join(_HERE, "dbscan_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(src2)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "clean", "all"])
    return _SO


_lib = None


def _load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(_SO)
        lib.tkref_sizeof_neigh.restype = ctypes.c_int
        lib.tkref_num_threads.restype = ctypes.c_int
        lib.tkref_distance.restype = ctypes.c_float
        lib.tkref_distance.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.tkref_init_rows.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]
        lib.tkref_trueknn.restype = ctypes.c_int
        lib.tkref_trueknn.argtypes = [
            ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_int,
            ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p,
            ctypes.c_void_p,
        ]
        lib.tkref_last_query_seconds.restype = ctypes.c_double
        lib.tkref_bruteforce.restype = ctypes.c_int
        lib.tkref_bruteforce.argtypes = [
            ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64,
            ctypes.c_void_p, ctypes.c_void_p,
        ]
        lib.dbref_dbscan.restype = ctypes.c_int
        lib.dbref_dbscan.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_int,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.dbref_dbscan_mt.restype = ctypes.c_int
        lib.dbref_dbscan_mt.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_int,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.dbref_dbscan_auto.restype = ctypes.c_int
        lib.dbref_dbscan_auto.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_int, ctypes.c_double,
                                          ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_void_p]
        lib.tkref_trueknn_rows.restype = ctypes.c_int
        lib.tkref_trueknn_rows.argtypes = lib.tkref_trueknn.argtypes
        lib.dbref_noise_count_mt.restype = ctypes.c_int64
        lib.dbref_noise_count_mt.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
        lib.dbref_ball_check.restype = ctypes.c_int64
        lib.dbref_ball_check.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_int, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        assert lib.tkref_sizeof_neigh() == NEIGH_DTYPE.itemsize
        _lib = lib
    return _lib


def num_threads():
    return int(_load().tkref_num_threads())


def _points(xyz):
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    if xyz.ndim != 2 or xyz.shape[1] not in (2, 3):
        raise ValueError("points must be (n,2) or (n,3)")
    if xyz.shape[1] == 2:  # hostCode.cpp:115-118
        xyz = np.ascontiguousarray(np.concatenate([xyz, np.zeros((len(xyz), 1), np.float32)], 1))
    return xyz


def distance(c_prim, org):
    a = np.ascontiguousarray(c_prim, dtype=np.float32)
    b = np.ascontiguousarray(org, dtype=np.float32)
    return float(_load().tkref_distance(a.ctypes.data, b.ctypes.data))


def trueknn(xyz, k, start_radius, order=ORDER_ASCENDING, seed=0, query_ids=None, max_rounds=64):
    """Run the restated solve.  Returns dict(fb, idx, dist, intersections, rounds, final_radius).

    ``fb`` is the reference's frameBuffer (n*k Neigh records) after the last round; ``idx`` /
    ``dist`` / ``intersections`` are views of it reshaped (n,k) / (n,k) / (n,).  With
    ``query_ids`` only those rows are solved (others keep their initial state).
    """
    lib = _load()
    xyz = _points(xyz)
    n = len(xyz)
    fb = np.zeros(n * k, dtype=NEIGH_DTYPE)
    lib.tkref_init_rows(fb.ctypes.data, n, k)
    q = None
    nq = n
    if query_ids is not None:
        q = np.ascontiguousarray(query_ids, dtype=np.int32)
        nq = len(q)
    fr = ctypes.c_float(0)
    rc = lib.tkref_trueknn(
        xyz.ctypes.data, n, k, ctypes.c_float(start_radius), order, seed,
        None if q is None else q.ctypes.data, nq, max_rounds, fb.ctypes.data, ctypes.byref(fr),
    )
    if rc < 0:
        raise OracleError({-1: "bad arguments", -2: "out of memory",
                           -3: "max_rounds reached with unfinished queries"}.get(rc, str(rc)))
    rows = fb.reshape(n, k)
    return {
        "fb": fb,
        "idx": rows["ind"],
        "dist": rows["dist"],
        "intersections": rows["intersections"][:, 0],
        "num_neighbors": rows["numNeighbors"][:, 0],
        "rounds": rc,
        "final_radius": float(fr.value),
        "query_seconds": float(lib.tkref_last_query_seconds()),
    }


def trueknn_rows(xyz, k, start_radius, query_ids, max_rounds=64):
    """The restated solve for sampled queries only, rows stored compactly (row t belongs to query_ids[t]): what
    ``trueknn(..., query_ids=q)`` gives in rows q, without the n*k frameBuffer (24 GB at 10^8 points, k = 10).
    Returns dict(idx (m,k), dist (m,k), intersections (m,), rounds, final_radius)."""
    lib = _load()
    xyz = _points(xyz)
    n = len(xyz)
    q = np.ascontiguousarray(query_ids, dtype=np.int32)
    rows = np.zeros(len(q) * k, dtype=NEIGH_DTYPE)
    lib.tkref_init_rows(rows.ctypes.data, len(q), k)
    fr = ctypes.c_float(0)
    rc = lib.tkref_trueknn_rows(xyz.ctypes.data, n, k, ctypes.c_float(start_radius), ORDER_ASCENDING, 0, q.ctypes.data, len(q),
                                max_rounds, rows.ctypes.data, ctypes.byref(fr))
    if rc < 0:
        raise OracleError({-1: "bad arguments", -2: "out of memory", -3: "max_rounds reached with unfinished queries"}.get(rc, str(rc)))
    rows = rows.reshape(len(q), k)
    return {"idx": rows["ind"], "dist": rows["dist"], "intersections": rows["intersections"][:, 0], "rounds": rc,
            "final_radius": float(fr.value), "query_seconds": float(lib.tkref_last_query_seconds())}


def trueknn_per_query(xyz, k, start_radii, max_rounds=64):
    """The opt-in per-query radius schedule (tknnSolveOptions.d_start_radii): row q is what the
    reference's loop (hostCode.cpp:285-340) produces for query q when it starts at start_radii[q] -- a row's result
    never depends on other rows, so the queries of each distinct start radius are simply run as one reference solve of
    their own.  Returns dict(idx, dist, intersections, rounds) with rounds = the largest number any query needed."""
    xyz = _points(xyz)
    n = len(xyz)
    radii = np.ascontiguousarray(start_radii, dtype=np.float32)
    if radii.shape != (n,):
        raise ValueError("one start radius per point")
    idx = np.empty((n, k), np.int32)
    dist = np.empty((n, k), np.float32)
    isect = np.empty(n, np.int64)
    rounds = 0
    for r in np.unique(radii):
        q = np.flatnonzero(radii == r).astype(np.int32)
        part = trueknn(xyz, k, float(r), query_ids=q, max_rounds=max_rounds)
        idx[q], dist[q], isect[q] = part["idx"][q], part["dist"][q], part["intersections"][q]
        rounds = max(rounds, part["rounds"])
    return {"idx": idx, "dist": dist, "intersections": isect, "rounds": rounds}


def bruteforce_knn(xyz, k, query_ids=None):
    lib = _load()
    xyz = _points(xyz)
    n = len(xyz)
    q = None
    nq = n
    if query_ids is not None:
        q = np.ascontiguousarray(query_ids, dtype=np.int32)
        nq = len(q)
    idx = np.empty((nq, k), np.int32)
    dist = np.empty((nq, k), np.float32)
    rc = lib.tkref_bruteforce(xyz.ctypes.data, n, k, None if q is None else q.ctypes.data, nq,
                              idx.ctypes.data, dist.ctypes.data)
    if rc:
        raise OracleError("bruteforce failed: %d" % rc)
    return idx, dist


def dbscan(xyz, eps, min_pts):
    """The RT-DBSCAN spec of oracle/dbscan_oracle.c: dict(labels, core, counts, clusters)."""
    lib = _load()
    xyz = _points(xyz)
    n = len(xyz)
    labels = np.empty(n, np.int32)
    core = np.empty(n, np.uint8)
    counts = np.empty(n, np.int32)
    rc = lib.dbref_dbscan(xyz.ctypes.data, n, ctypes.c_float(eps), int(min_pts), labels.ctypes.data,
                          core.ctypes.data, counts.ctypes.data)
    if rc < 0:
        raise OracleError("dbscan oracle failed: %d" % rc)
    return {"labels": labels, "core": core.astype(bool), "counts": counts, "clusters": rc}


def dbscan_threaded(xyz, eps, min_pts):
    """The same spec on all host cores (dbref_dbscan_mt): what bench.py times as RT-DBSCAN's CPU
    baseline.  dict(labels, core, clusters, seconds=[grid, core flags, unions, labels])."""
    lib = _load()
    xyz = _points(xyz)
    n = len(xyz)
    labels = np.empty(n, np.int32)
    core = np.empty(n, np.uint8)
    seconds = np.zeros(4, np.float64)
    rc = lib.dbref_dbscan_mt(xyz.ctypes.data, n, ctypes.c_float(eps), int(min_pts), labels.ctypes.data,
                             core.ctypes.data, seconds.ctypes.data)
    if rc < 0:
        raise OracleError("threaded dbscan failed: %d" % rc)
    return {"labels": labels, "core": core.astype(bool), "clusters": rc, "seconds": seconds}


def dbscan_auto(xyz, eps0, min_pts, max_noise=0.05, max_rounds=32):
    """The "eps auto-grown" spec of oracle/dbscan_oracle.c (dbref_dbscan_auto): eps doubles from eps0 until at most
    floor(max_noise * n) points are noise.  dict(labels, core, clusters, eps, rounds, noise)."""
    lib = _load()
    xyz = _points(xyz)
    n = len(xyz)
    labels = np.empty(n, np.int32)
    core = np.empty(n, np.uint8)
    eps = ctypes.c_float(0)
    rounds = ctypes.c_int(0)
    noise = ctypes.c_int64(0)
    rc = lib.dbref_dbscan_auto(xyz.ctypes.data, n, ctypes.c_float(eps0), int(min_pts), ctypes.c_double(max_noise), int(max_rounds),
                               labels.ctypes.data, core.ctypes.data, ctypes.byref(eps), ctypes.byref(rounds), ctypes.byref(noise))
    if rc < 0:
        raise OracleError("dbscan_auto failed: %d%s" % (rc, " (max_rounds exhausted)" if rc == -3 else ""))
    return {"labels": labels, "core": core.astype(bool), "clusters": rc, "eps": float(eps.value), "rounds": int(rounds.value),
            "noise": int(noise.value)}



def dbscan_noise_count(xyz, eps, min_pts, want_core=False):
    """Number of points DBSCAN(eps, min_pts) labels -1, by the spec, without building clusters (all host cores)."""
    lib = _load()
    xyz = _points(xyz)
    core = np.zeros(len(xyz), np.uint8) if want_core else None
    rc = int(lib.dbref_noise_count_mt(xyz.ctypes.data, len(xyz), ctypes.c_float(eps), int(min_pts), None if core is None else core.ctypes.data))
    if rc < 0:
        raise OracleError("dbref_noise_count_mt: %d" % rc)
    return (rc, core.astype(bool)) if want_core else rc


def dbscan_auto_counts(xyz, eps0, min_pts, max_noise, max_rounds=32):
    """The growth loop of dbref_dbscan_auto with count-only rounds: dict(rounds, eps, noise) of the round that brings the
    noise under floor(max_noise * n) -- what the full spec reports, at a cost that allows BASELINE config 5's 50 M points."""
    xyz = _points(xyz)
    bound = int(np.floor(float(max_noise) * len(xyz)))
    eps = np.float32(eps0)
    for t in range(int(max_rounds)):
        noise = dbscan_noise_count(xyz, float(eps), min_pts)
        if noise <= bound:
            return {"rounds": t + 1, "eps": float(eps), "noise": int(noise)}
        eps = np.float32(eps * np.float32(2))  # hostCode.cpp:321
    raise OracleError("max_rounds doublings of eps did not bring the noise under the bound")


def dbscan_ball_check(xyz, eps, min_pts, labels, core, sample):
    """Recompute the eps-balls of the sampled points with the spec's arithmetic and check core flags and labels of a
    finished clustering on them (dbref_ball_check).  Returns dict(violations, first_bad)."""
    lib = _load()
    xyz = _points(xyz)
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    core = np.ascontiguousarray(core).astype(np.uint8)
    sample = np.ascontiguousarray(sample, dtype=np.int32)
    if len(labels) != len(xyz) or len(core) != len(xyz):
        raise ValueError("labels / core flags for every point")
    first = ctypes.c_int32(-1)
    rc = int(lib.dbref_ball_check(xyz.ctypes.data, len(xyz), ctypes.c_float(eps), int(min_pts), labels.ctypes.data, core.ctypes.data,
                                  sample.ctypes.data, len(sample), ctypes.byref(first)))
    if rc < 0:
        raise OracleError("dbref_ball_check: %d" % rc)
    return {"violations": rc, "first_bad": int(first.value)}
