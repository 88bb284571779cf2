"""Start-radius helper: counterpart of the reference's samples/s01-trueknn/Util/random_sample.py.

That script draws 100 random points of the dataset, runs a nearest-neighbour search among the
sample (sklearn ball tree) and prints the smallest neighbour distance as the start radius
(random_sample.py:10-31; the README's "Random Sampling For Start Radius").  As shipped it asks for
n_neighbors=1 and then indexes columns 1..3 of the result, which would raise before printing; the
intent -- the minimum distance from a sampled point to another sampled point -- is what this module
computes, seeded and without sklearn.  Host-side numpy; nothing here touches the GPU.
"""
import numpy as np


def sample_start_radius(points, n_samples=100, seed=0):
    """min over a seeded sample of the distance to the nearest *other* sampled point (float32)."""
    pts = np.asarray(points, dtype=np.float32)
    if pts.ndim != 2 or pts.shape[1] not in (2, 3):
        raise ValueError("points must be (n,2) or (n,3)")
    n = len(pts)
    if n < 2:
        raise ValueError("need at least two points")
    rng = np.random.default_rng(seed)
    pick = rng.choice(n, size=min(n_samples, n), replace=False)
    s = pts[pick].astype(np.float64)
    d2 = ((s[:, None, :] - s[None, :, :]) ** 2).sum(-1)
    np.fill_diagonal(d2, np.inf)
    best = float(np.sqrt(d2.min()))
    if best == 0.0:  # duplicates in the sample: fall back to the smallest positive distance
        pos = d2[d2 > 0]
        best = float(np.sqrt(pos.min())) if pos.size else 0.0
    return best
