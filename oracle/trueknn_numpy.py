"""Independent numpy/scipy restatement of the TrueKNN result function.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED (see trueknn_oracle.c).

Where ``trueknn_oracle.c`` replays the reference's programs call by call
(samples/s01-trueknn/deviceCode.cu:62-153 inside the round loop of hostCode.cpp:285-340), this
file states what that replay must come to, with none of its machinery:

  * level t uses radius r_t = fl32(r_{t-1} * 2), r_0 = start radius           (hostCode.cpp:321)
  * candidates of query q at level t: every p (q itself included) with
    fl32(c_p - r_t) <= q <= fl32(c_p + r_t) on all three axes                 (deviceCode.cu:38-56)
  * q finishes at the first level t* at which at least k candidates other than q exist; its row
    is the k smallest (distance, first level, index) triples among those candidates, "first level"
    being the level at which a candidate first was one: the reference's lists persist over rounds
    and a new entry goes BEHIND listed ones of equal distance (deviceCode.cu:77-85 skip,
    :116-134 strict '<'; ascending-index visit order inside a round)
    distance = sqrt((dx*dx + dy*dy) + dz*dz) in float32, each operation rounded, as
    deviceCode.cu:110-113 is written
  * intersections(q) = sum of candidate counts (self included) over levels 0..t*   (deviceCode.cu:74)
  * rounds = 1 + max t*                                                        (hostCode.cpp:285-340)

Candidates are proposed by a Chebyshev (p = inf) ball query of scipy's cKDTree with a small
margin; the fp32 box test above decides.  The two files share no code, so agreement between them
is evidence that the restatement is self-consistent -- it is not reference output.
"""
import numpy as np
from scipy.spatial import cKDTree


def distance32(c_prim, org):
    """sqrt((dx*dx + dy*dy) + dz*dz) with every operation rounded to float32 (numpy float32
    arithmetic rounds each operation; no fused multiply-add)."""
    d = c_prim.astype(np.float32) - org.astype(np.float32)
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    return np.sqrt(((x * x) + (y * y)) + (z * z), dtype=np.float32)


def trueknn_numpy(xyz, k, start_radius, max_rounds=64, query_ids=None, stop_quietly=False, ids=None):
    """``ids`` (optional): identity of each point for the self test and the tie order (defaults to
    the position); ``stop_quietly``: at max_rounds return with level -1 for unfinished queries."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    if xyz.shape[1] == 2:
        xyz = np.concatenate([xyz, np.zeros((len(xyz), 1), np.float32)], 1)
    n = len(xyz)
    queries = np.arange(n) if query_ids is None else np.asarray(query_ids)
    idx = np.full((n, k), -1, np.int32)
    dist = np.full((n, k), np.float32(3.402823466e38), np.float32)
    isect = np.zeros(n, np.int64)
    level_of = np.full(n, -1, np.int32)
    tree = cKDTree(xyz.astype(np.float64))
    radius = np.float32(start_radius)
    active = queries.copy()
    rounds = 0
    first_seen = {int(q): {} for q in active}  # query -> {candidate position: level at which it first was one}
    while len(active):
        if rounds >= max_rounds:
            if stop_quietly:
                break
            raise RuntimeError("max_rounds reached with unfinished queries")
        r = np.float32(radius)
        reach = abs(float(r)) * 1.0001 + 1e-30 + 1e-6 * float(np.abs(xyz).max())
        proposals = tree.query_ball_point(xyz[active].astype(np.float64), reach, p=np.inf)
        still = []
        for q, prop in zip(active, proposals):
            p = np.asarray(prop, dtype=np.int64)
            c = xyz[p]
            m, s = (c - r).astype(np.float32), (c + r).astype(np.float32)
            lo, hi = np.minimum(m, s), np.maximum(m, s)
            inside = np.all((lo <= xyz[q]) & (xyz[q] <= hi), axis=1)
            p = p[inside]
            isect[q] += len(p)
            others = p[p != q] if ids is None else p[ids[p] != ids[q]]
            seen = first_seen[int(q)]
            for c_pos in others:
                seen.setdefault(int(c_pos), rounds)
            if len(others) >= k:
                d = distance32(xyz[others], xyz[q])
                names = others if ids is None else ids[others]
                since = np.asarray([seen[int(c_pos)] for c_pos in others], np.int64)
                order = np.lexsort((names, since, d))[:k]
                idx[q] = names[order]
                dist[q] = d[order]
                level_of[q] = rounds
                del first_seen[int(q)]
            else:
                still.append(q)
        rounds += 1
        active = np.asarray(still, dtype=np.int64)
        if len(active):
            radius = np.float32(radius * np.float32(2))
    return {"idx": idx, "dist": dist, "intersections": isect, "rounds": rounds,
            "final_radius": float(radius), "level": level_of}
