"""Synthetic inputs for the TrueKNN / RT-DBSCAN path and the reference's CSV input format.

The reference's sample dataset is absent from its tree (``.MISSING_LARGE_BLOBS``), so every
benchmark and test input is generated here; the recipes are the ones SURVEY.md section 8(d) fixes.
Host-side numpy only -- nothing here touches the GPU.
"""
import io

import numpy as np

CHUNK = 1 << 20  # counter-based generation granule: point i always comes from chunk i // CHUNK


def start_radius(n, k, extent=1.0):
    """r0 = 0.25 * (k/n)^(1/3) on unit-cube data: expected box population k/8."""
    return float(0.25 * extent * (float(k) / float(n)) ** (1.0 / 3.0))


def uniform3d(n, seed=0):
    """C1 / C2: ``default_rng(seed).random((n, 3), float32)``, uniform in [0,1)^3."""
    return np.random.default_rng(seed).random((n, 3), dtype=np.float32)


def uniform3d_counter(start, stop, seed=0):
    """C4: rows [start, stop) of a point set defined independently of how it is sharded.

    Chunk c (rows c*CHUNK ...) is drawn from Philox(key=seed) jumped c times, so any rank can
    generate any slice and 1/2/4/8-GPU runs see the same 100 M points.
    """
    out = np.empty((stop - start, 3), np.float32)
    c0, c1 = start // CHUNK, (stop - 1) // CHUNK if stop > start else start // CHUNK
    for c in range(c0, c1 + 1):
        lo, hi = max(start, c * CHUNK), min(stop, (c + 1) * CHUNK)
        if hi <= lo:
            continue
        bitgen = np.random.Philox(key=seed).jumped(c)
        block = np.random.Generator(bitgen).random((CHUNK, 3), dtype=np.float32)
        out[lo - start:hi - start] = block[lo - c * CHUNK:hi - c * CHUNK]
    return out


def gaussian_mixture3d(n, components=64, sigma=0.02, seed=1):
    """C3: equal-weight isotropic mixture, means uniform in [0,1)^3."""
    rng = np.random.default_rng(seed)
    means = rng.random((components, 3))
    which = rng.integers(0, components, n)
    pts = means[which] + rng.normal(0.0, sigma, (n, 3))
    return pts.astype(np.float32)


def taxi_like2d(n, components=256, dup_fraction=0.05, seed=2):
    """C5: heavy-tailed 2-D mixture (log-normal sigmas) with exact duplicates; returned (n,2)."""
    rng = np.random.default_rng(seed)
    means = rng.random((components, 2))
    sig = np.exp(rng.normal(np.log(0.01), 1.0, components))
    weights = rng.pareto(1.5, components) + 1e-3
    weights /= weights.sum()
    which = rng.choice(components, n, p=weights)
    pts = (means[which] + rng.normal(0.0, 1.0, (n, 2)) * sig[which, None]).astype(np.float32)
    ndup = int(n * dup_fraction)
    if ndup:
        dst = rng.choice(n, ndup, replace=False)
        src = rng.integers(0, n, ndup)
        pts[dst] = pts[src]
    return pts


def pad_to_3d(points):
    """2-D input gets z = 0 (samples/s01-trueknn/hostCode.cpp:115-118)."""
    points = np.ascontiguousarray(points, dtype=np.float32)
    if points.ndim != 2 or points.shape[1] not in (2, 3):
        raise ValueError("points must have shape (n,2) or (n,3), got %r" % (points.shape,))
    if points.shape[1] == 3:
        return points
    return np.ascontiguousarray(
        np.concatenate([points, np.zeros((len(points), 1), np.float32)], axis=1))


def read_csv_points(path_or_text, n_points, dim):
    """Parse the reference's input format (samples/s01-trueknn/hostCode.cpp:83-104).

    Per line, numbers are read with ``stream >> float`` and a single ',' after a number is
    skipped; anything else ends the line.  The reader checks ``count > 0`` only between lines
    (``while (getline && count > 0)``), so the line on which the budget of ``n_points*dim``
    numbers runs out is still consumed to its end, exactly as in the reference.  Returns the flat
    float32 vector reshaped to (len // dim, dim) -- the reference then builds one sphere per
    ``dim`` numbers (hostCode.cpp:115-124).
    """
    if dim not in (2, 3):
        raise ValueError("dim must be 2 or 3 (hostCode.cpp:115-124 handles nothing else)")
    if isinstance(path_or_text, (str, bytes)) and "\n" not in str(path_or_text) and \
            "," not in str(path_or_text):
        fh = open(path_or_text, "r")
    else:
        fh = io.StringIO(path_or_text if isinstance(path_or_text, str) else path_or_text.decode())
    vals = []
    count = int(n_points) * dim
    with fh:
        for line in fh:
            if count <= 0:
                break
            pos, ln = 0, line.rstrip("\n")
            while True:
                # operator>> skips leading whitespace, then parses the longest float prefix
                while pos < len(ln) and ln[pos] in " \t\r\v\f":
                    pos += 1
                end = _float_prefix(ln, pos)
                if end == pos:
                    break
                vals.append(float(ln[pos:end]))
                count -= 1
                pos = end
                if pos < len(ln) and ln[pos] == ",":
                    pos += 1
    flat = np.asarray(vals, dtype=np.float32)
    usable = (len(flat) // dim) * dim
    return flat[:usable].reshape(-1, dim)


def _float_prefix(s, pos):
    """End index of the longest prefix of s[pos:] that ``istream >> float`` would accept."""
    i, n = pos, len(s)
    if i < n and s[i] in "+-":
        i += 1
    digits = 0
    while i < n and s[i].isdigit():
        i += 1
        digits += 1
    if i < n and s[i] == ".":
        i += 1
        while i < n and s[i].isdigit():
            i += 1
            digits += 1
    if digits == 0:
        return pos
    if i < n and s[i] in "eE":
        j = i + 1
        if j < n and s[j] in "+-":
            j += 1
        if j < n and s[j].isdigit():
            while j < n and s[j].isdigit():
                j += 1
            i = j
    return i


def write_csv_points(path, points):
    """Write ``x,y[,z]`` lines readable by the reference sample and by read_csv_points."""
    pts = np.asarray(points, dtype=np.float32)
    with open(path, "w") as fh:
        for row in pts:
            fh.write(",".join(repr(float(v)) for v in row) + "\n")


def boundary_band(anchors: int, r0: float, seed: int = 0, levels: int = 3, spread: float = 600.0) -> np.ndarray:
    """Points that sit ON the candidate-box faces of other points, a few ulps either side.

    Around each of ``anchors`` points (coordinates up to +-``spread``, i.e. far larger than r0) every
    axis, sign and level j < ``levels`` gets satellites at distance r0*2^j, nudged by -2..+2 ulps
    along that axis; the other two coordinates stay well inside the box.  Whether such a point is a
    candidate depends on the rounding of fl(c - r) and fl(c + r) (deviceCode.cu:38-56), which is
    what tests of the box-test arithmetic want to see exercised at every magnitude.
    """
    rng = np.random.default_rng(seed)
    mags = np.array([1e-3, 0.07, 1.0, 37.5, spread], dtype=np.float64)
    out = []
    for a in range(anchors):
        c = ((rng.random(3) * 2 - 1) * mags[a % len(mags)]).astype(np.float32)
        out.append(c)
        for j in range(levels):
            r = np.float32(r0) * np.float32(2.0 ** j)
            for ax in range(3):
                for sign in (-1.0, 1.0):
                    for nudge in range(-2, 3):
                        p = c + ((rng.random(3) - 0.5) * float(r0) * 0.5).astype(np.float32)
                        v = np.float32(c[ax] + np.float32(sign) * r)
                        step = np.float32(np.inf if nudge > 0 else -np.inf)
                        for _ in range(abs(nudge)):
                            v = np.nextafter(v, step)
                        p[ax] = v
                        out.append(p)
    pts = np.asarray(out, dtype=np.float32)
    return pts[rng.permutation(len(pts))]


def cross_round_ties(clumps: int = 150, seed: int = 17) -> np.ndarray:
    """Isolated clumps q, A, B, C with |qA| = |qB| = |qC| = 1.125 exactly (a 1-2-2 triple scaled by
    3/8): A = q + (.375, .75, .75) lies inside q's box of radius 1, B = q + (1.125, 0, 0) and
    C = q - (0, 0, 1.125) only inside the box of radius 2.  With start radius 1 and k = 2 the query q
    sees one other in round 0 (A) and three in round 1; the reference's persistent list keeps A first
    and then the smaller index of B, C -- whatever A's index is.  Indices are shuffled so that every
    order occurs."""
    rng = np.random.default_rng(seed)
    side = int(np.ceil(clumps ** (1 / 3)))
    pts = []
    for c in range(clumps):
        centre = np.array([c % side, (c // side) % side, c // (side * side)], np.float32) * np.float32(16)
        pts += [centre, centre + np.float32([0.375, 0.75, 0.75]), centre + np.float32([1.125, 0, 0]),
                centre - np.float32([0, 0, 1.125])]
    pts = np.asarray(pts, np.float32)
    return pts[rng.permutation(len(pts))]
