"""The sharded (Morton tiles + halo exchange) path against the single-process CPU checker."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
from owlraytracing_amd import datasets

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_worker.py")


def _run(engine, world, n, k, name, tmp_path, port, env_extra=None, timeout=600):
    out = str(tmp_path / ("rows_%s_%d.npz" % (name, world)))
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "OMP_NUM_THREADS": "2"})
    env.update(env_extra or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), WORKER, engine, str(n), str(k), name, out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:] + r.stderr[-3000:])
    print("\n".join(l for l in r.stdout.splitlines() if l.startswith("rank ")))
    return np.load(out)


def _check(got, name, n, k, r0=None):
    from dist_worker import make_points
    pts = make_points(name, n)
    ref = oracle.trueknn(pts, k, datasets.start_radius(n, k) if r0 is None else r0)
    assert np.array_equal(got["gids"], np.arange(n))
    assert np.array_equal(got["idx"], ref["idx"])
    assert np.array_equal(got["dist"], ref["dist"])
    assert np.array_equal(got["isect"], ref["intersections"])
    assert int(got["rounds"]) == ref["rounds"]


@pytest.mark.parametrize("count", [1, 2, 1023, 1024, 1025, (1 << 24) + 1, 33_554_436, 50_000_000, (1 << 31) + 7])
def test_splitter_sample_positions_stay_inside_large_tiles(count):
    """The splitter sample of load_points indexes the rank's sorted codes: positions must be exact
    integers inside [0, count-1] at any tile size (a float32 linspace returns `count` itself from
    2**24 points on: 33 554 436 and 50 000 000 are the cases the round-1 review measured)."""
    import torch

    from owlraytracing_amd.distributed import sample_positions
    pick = sample_positions(count, 1024, torch.device("cpu"))
    assert pick.dtype == torch.int64 and len(pick) == 1024
    assert int(pick[0]) == 0 and int(pick[-1]) == count - 1
    assert bool((pick[1:] >= pick[:-1]).all()) and int(pick.max()) < count
    # evenly spaced: neighbouring gaps differ by at most one
    gaps = pick[1:] - pick[:-1]
    assert int(gaps.max() - gaps.min()) <= 1


@pytest.mark.parametrize("world,name,n,k", [(2, "uniform", 3000, 5), (3, "clustered", 2500, 4), (2, "planar", 2000, 3)])
def test_tiles_and_halo_exchange_reproduce_the_single_process_result(tmp_path, world, name, n, k):
    """gloo, CPU ranks, checker-backed engine: partition, halo selection, straggler loop, global ids."""
    got = _run("checker", world, n, k, name, tmp_path, 29611 + world)
    _check(got, name, n, k)
    assert int(got["halo_points"]) > 0


@pytest.mark.parametrize("world", [2, 3])
def test_strong_scaling_split_of_the_counter_based_set_equals_one_rank(tmp_path, world):
    """`bench.py --scaling strong`: ONE point set (counter-based generator, the same points whatever the number of
    ranks) cut into `world` Morton tiles; rows, intersection counts and rounds equal the single-process result."""
    n, k = 2 * 1024 + 77, 4  # crosses nothing special; datasets.CHUNK-sized sets are the GPU-side runs
    got = _run("checker", world, n, k, "counter", tmp_path, 29680 + world)
    _check(got, "counter", n, k)


def test_stragglers_force_a_wider_halo(tmp_path):
    # a start radius far too small: the first halo (1 level) cannot serve the final radius level
    got = _run("checker", 2, 1500, 6, "uniform", tmp_path, 29631, {"START_RADIUS": "0.004", "HALO_LEVELS": "1"})
    _check(got, "uniform", 1500, 6, r0=0.004)
    assert int(got["exchanges"]) > 1


def test_straggler_rounds_send_the_shell_and_solve_only_the_stragglers(tmp_path):
    """a widening round re-sent the whole halo at the doubled radius and re-solved every query.  Now only
    the points between the two radii travel (the shell) and only the unfinished queries are solved again
    (tknnSolveOptions.phase = 3): with HALO_LEVELS=0 on the clustered set every level is a round of its own -- rows,
    intersection counts and rounds still equal the single-process result, each later exchange carries the shell alone
    (what the peers hold is never sent twice: the shells add up to exactly the halo the final radius selects), and
    the sum of the per-query intersection counts the driver reports is the reference's."""
    n, k = 2500, 4
    got = _run("checker", 3, n, k, "clustered", tmp_path, 29634, {"HALO_LEVELS": "0", "START_RADIUS": "0.01"})
    _check(got, "clustered", n, k, r0=0.01)
    by = got["halo_by_exchange"]
    assert int(got["exchanges"]) == len(by) >= 3
    assert int(by.sum()) == int(got["halo_points"])  # rank 0's halo tree = the first halo + the shells, nothing twice
    # a whole re-send would carry at least the previous halo again: every shell is smaller than what is held already
    assert all(int(by[j]) < int(by[:j].sum()) for j in range(2, len(by)))
    from dist_worker import make_points
    ref = oracle.trueknn(make_points("clustered", n), k, 0.01)
    assert int(got["isect"].sum()) == int(ref["intersections"].sum())


def test_fixed_capacity_exchange_carries_counts_in_the_first_row(tmp_path):
    """_Comm.exchange_rows with a tag: after a first (counts-first) exchange under the tag, messages have a fixed capacity
    and their first row holds the count -- one exchange instead of two; a pair that outgrows its capacity falls back to
    exact-size messages.  Two gloo ranks trade row blocks of changing sizes; every block arrives intact."""
    script = tmp_path / "xchg.py"
    script.write_text("""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from owlraytracing_amd.distributed import _Comm
dist.init_process_group("gloo")
c = _Comm()
r, w = c.rank, c.world
dev = torch.device("cpu")
def block(src, dst, m, step):
    return (torch.arange(m * 4, dtype=torch.float32).reshape(m, 4) + 1000.0 * src + 10.0 * dst + 0.25 * step)
sizes = [(5, 7), (5, 7), (6, 3), (40, 2), (0, 0), (300, 300)]   # rows 0 -> 1, rows 1 -> 0 per step: same, shrinking, outgrowing, empty
for step, (a, b) in enumerate(sizes):
    m_out = [0, 0]
    m_out[1 - r] = a if r == 0 else b
    blocks = [block(r, p, m_out[p], step) for p in range(w)]
    got = c.exchange_rows(blocks, 4, torch.float32, dev, tag="t")
    m_in = b if r == 0 else a
    want = block(1 - r, r, m_in, step)
    assert got[1 - r].shape == want.shape and torch.equal(got[1 - r], want), (step, r, got[1 - r].shape, want.shape)
    if step >= 1:
        assert ("t", "in") in c._caps
print("rank %%d ok caps %%s" %% (r, c._caps[("t", "in")]))
dist.destroy_process_group()
""" % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29637", str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count(" ok caps ") == 2


@pytest.mark.gpu
def test_straggler_rounds_with_the_hip_engine(tmp_path):
    """The same with the real engine (tknnSolveOptions.phase = 3 over own + widened halo tree), three ranks sharing the GPU."""
    n, k = 150_000, 8
    got = _run("hip", 3, n, k, "clustered", tmp_path, 29648, {"HALO_LEVELS": "0", "START_RADIUS": "0.001"})
    _check(got, "clustered", n, k, r0=0.001)
    by = got["halo_by_exchange"]
    assert len(by) >= 3 and int(by.sum()) == int(got["halo_points"])


@pytest.mark.gpu
def test_two_ranks_sharing_one_gpu_with_the_hip_engine(tmp_path):
    """Real engine (ids, halo tree, allow_unfinished) under the sharded driver; messages host-staged
    through gloo because both ranks sit on the one GPU of the box."""
    n, k = 200_000, 10
    got = _run("hip", 2, n, k, "uniform", tmp_path, 29641, {"HALO_LEVELS": "2"})
    _check(got, "uniform", n, k)
    # the worker solves twice and reports the second solve: its halo rows were selected in ONE pass straight into messages of
    # the capacity the first exchange taught both ends (tknnHaloSelectFixed), counts read with the headers
    assert bool(got["one_pass"])


@pytest.mark.gpu
def test_interior_and_boundary_phases_reproduce_the_single_process_result(tmp_path):
    """TKNN_SHARD_OVERLAP=1: interior queries solved in the own tree on a side stream and host thread while the halo
    travels, boundary queries afterwards with own + halo tree (tknnSolveOptions.phase 1 / 2): the same rows."""
    n, k = 300_000, 10
    got = _run("hip", 3, n, k, "uniform", tmp_path, 29645, {"HALO_LEVELS": "2", "TKNN_SHARD_OVERLAP": "1"})
    _check(got, "uniform", n, k)
    got = _run("hip", 2, 120_000, 6, "clustered", tmp_path, 29646, {"HALO_LEVELS": "1", "TKNN_SHARD_OVERLAP": "1", "START_RADIUS": "0.002"})
    _check(got, "clustered", 120_000, 6, r0=0.002)


@pytest.mark.gpu
def test_four_ranks_sharing_one_gpu_cross_z_curve_jumps(tmp_path):
    """4 Morton tiles: two of the three tile boundaries sit at jumps of the Z curve; the per-cell
    tile boxes keep the halo small there (a single bounding box per tile would pull in whole tiles)."""
    n, k = 400_000, 10
    got = _run("hip", 4, n, k, "uniform", tmp_path, 29651, {"HALO_LEVELS": "2"})
    _check(got, "uniform", n, k)
    assert bool(got["one_pass"])
    assert int(got["halo_points"]) < int(got["tile"]) // 2


def _check_dbscan(got, name, n, eps, min_pts):
    from dist_worker import make_points
    ref = oracle.dbscan(make_points(name, n), eps, min_pts)
    assert np.array_equal(got["gids"], np.arange(n))
    assert np.array_equal(got["core"], ref["core"].astype(bool))
    assert int(got["clusters"]) == int(ref["clusters"])
    assert np.array_equal(got["labels"], ref["labels"])


@pytest.mark.parametrize("world,name,n,eps,min_pts", [(2, "clustered", 6000, 0.02, 5), (3, "uniform", 5000, 0.035, 4),
                                                     (3, "planar", 6000, 0.004, 6)])
def test_sharded_dbscan_equals_the_single_process_spec(tmp_path, world, name, n, eps, min_pts):
    """Tiles + 2-eps halo + label propagation between ranks (gloo, CPU stand-in engine running the
    spec per tile): labels, core flags and cluster count equal oracle.dbscan over the whole set."""
    eps = float(np.float32(eps))
    got = _run("checker", world, n, 4, name, tmp_path, 29660 + world, {"DBSCAN_EPS": repr(eps), "DBSCAN_MINPTS": str(min_pts)})
    _check_dbscan(got, name, n, eps, min_pts)


def test_sharded_dbscan_with_auto_grown_eps_equals_the_spec(tmp_path):
    """BASELINE config 5's rule over tiles: eps doubles until the noise share of the WHOLE set is under the bound."""
    from dist_worker import make_points
    n, eps0, min_pts, max_noise = 6000, float(np.float32(0.001)), 5, 0.08
    got = _run("checker", 3, n, 4, "planar", tmp_path, 29675, {"DBSCAN_EPS": repr(eps0), "DBSCAN_MINPTS": str(min_pts),
                                                              "DBSCAN_MAX_NOISE": repr(max_noise)})
    ref = oracle.dbscan_auto(make_points("planar", n), eps0, min_pts, max_noise)
    assert ref["rounds"] > 1, "the start value must be too small for the test to mean anything"
    assert int(got["eps_rounds"]) == ref["rounds"] and float(got["eps"]) == ref["eps"] and int(got["noise"]) == ref["noise"]
    assert np.array_equal(got["labels"], ref["labels"]) and np.array_equal(got["core"], ref["core"])


@pytest.mark.gpu
def test_sharded_dbscan_auto_with_the_hip_engine(tmp_path):
    from dist_worker import make_points
    n, eps0, min_pts, max_noise = 200_000, float(np.float32(0.0015)), 5, 0.03
    got = _run("hip", 3, n, 4, "clustered", tmp_path, 29676, {"DBSCAN_EPS": repr(eps0), "DBSCAN_MINPTS": str(min_pts),
                                                               "DBSCAN_MAX_NOISE": repr(max_noise)})
    ref = oracle.dbscan_auto(make_points("clustered", n), eps0, min_pts, max_noise)
    assert ref["rounds"] > 1
    assert int(got["eps_rounds"]) == ref["rounds"] and float(got["eps"]) == ref["eps"] and int(got["noise"]) == ref["noise"]
    assert np.array_equal(got["labels"], ref["labels"]) and np.array_equal(got["core"], ref["core"])


@pytest.mark.gpu
def test_sharded_dbscan_with_the_hip_engine(tmp_path):
    n, eps, min_pts = 300_000, float(np.float32(0.012)), 5
    got = _run("hip", 3, n, 4, "clustered", tmp_path, 29671, {"DBSCAN_EPS": repr(eps), "DBSCAN_MINPTS": str(min_pts)})
    _check_dbscan(got, "clustered", n, eps, min_pts)


@pytest.mark.gpu
def test_halo_select_matches_a_plain_selection():
    """tknnHaloSelect (send side of the exchange): per peer, exactly the points inside any of its boxes,
    each once, as wire rows x y z id-bits."""
    import torch

    from owlraytracing_amd import datasets
    from owlraytracing_amd.trueknn import TrueKNN
    n, npeers = 150_000, 5
    pts = datasets.uniform3d(n, seed=31)
    pts[7] = np.nan  # never selected
    ids = (np.arange(n, dtype=np.int32) * 3 + 11)
    rng = np.random.default_rng(32)
    lo = rng.random((40, 3)).astype(np.float32) * 0.9
    hi = lo + rng.random((40, 3)).astype(np.float32) * 0.15
    hi[0] = lo[0]  # a degenerate box
    lo[1], hi[1] = pts[100], pts[100]  # a box that is exactly one point: closed boundaries
    peer = rng.integers(0, npeers, 40).astype(np.int32)
    peer[peer == 2] = 3  # a peer without boxes
    eng = TrueKNN()
    eng.build(torch.from_numpy(pts).cuda(), torch.from_numpy(ids).cuda())
    rows, counts = eng.halo_select(torch.from_numpy(np.concatenate([lo, hi], axis=1)), torch.from_numpy(peer), npeers)
    rows = rows.cpu()
    assert len(counts) == npeers and sum(counts) == len(rows) and counts[2] == 0
    start = 0
    for p in range(npeers):
        seg = rows[start:start + counts[p]]
        start += counts[p]
        got_ids = np.sort(seg[:, 3].contiguous().view(torch.int32).numpy())
        inside = np.zeros(n, bool)
        for j in np.nonzero(peer == p)[0]:
            inside |= np.all((pts >= lo[j]) & (pts <= hi[j]), axis=1)
        assert np.array_equal(got_ids, np.sort(ids[inside])), p
        # coordinates travel with their ids
        back = {int(i): tuple(r) for i, r in zip(seg[:, 3].contiguous().view(torch.int32).numpy(), seg[:, :3].numpy())}
        for i in list(back)[:50]:
            assert back[i] == tuple(pts[(i - 11) // 3])
    assert ids[100] in rows[:, 3].contiguous().view(torch.int32).numpy()
    eng.close()


@pytest.mark.gpu
def test_rccl_branch_with_one_rank(tmp_path):
    """every multi-rank test initialised gloo, so the branch a real multi-GPU run takes -- backend nccl (= RCCL):
    `init_process_group("nccl", device_id=...)`, DEVICE tensors through all_reduce / all_gather / batch_isend_irecv, the
    fixed-capacity exchange's no-peer path -- had never run inside the suite.  One rank is what a one-GPU box allows (RCCL
    refuses two ranks on one device); TrueKNN with a straggler round, RT-DBSCAN and the auto-eps loop go through it."""
    from dist_worker import make_points
    nccl = {"DIST_BACKEND": "nccl"}
    n, k = 150_000, 10
    got = _run("hip", 1, n, k, "uniform", tmp_path, 29681, dict(nccl, HALO_LEVELS="1"))
    _check(got, "uniform", n, k)
    got = _run("hip", 1, 80_000, 6, "clustered", tmp_path, 29682, dict(nccl, HALO_LEVELS="0", START_RADIUS="0.001"))
    _check(got, "clustered", 80_000, 6, r0=0.001)
    assert len(got["halo_by_exchange"]) >= 2  # (stragglers: more than one round of the all-reduce)
    n, eps, min_pts = 120_000, float(np.float32(0.012)), 5
    got = _run("hip", 1, n, 4, "clustered", tmp_path, 29683, dict(nccl, DBSCAN_EPS=repr(eps), DBSCAN_MINPTS=str(min_pts)))
    _check_dbscan(got, "clustered", n, eps, min_pts)
    eps0, max_noise = float(np.float32(0.0015)), 0.03
    got = _run("hip", 1, n, 4, "clustered", tmp_path, 29684, dict(nccl, DBSCAN_EPS=repr(eps0), DBSCAN_MINPTS=str(min_pts), DBSCAN_MAX_NOISE=repr(max_noise)))
    ref = oracle.dbscan_auto(make_points("clustered", n), eps0, min_pts, max_noise)
    assert int(got["eps_rounds"]) == ref["rounds"] and float(got["eps"]) == ref["eps"] and int(got["noise"]) == ref["noise"]
    assert np.array_equal(got["labels"], ref["labels"]) and np.array_equal(got["core"], ref["core"])


@pytest.mark.gpu
def test_config4_shaped_tiles_at_size(tmp_path):
    """BASELINE config 4's tiling had only run at <= 400 k points.  Here the counter-based uniform set (what
    `bench.py --gpus N` loads) at 20 M points is cut into 4 Morton tiles of 5 M -- ranks sharing the one GPU, host-staged
    messages --, solved twice (the second solve selects its halo in one pass into fixed-capacity messages), and the rows of every
    97th id are compared with the CPU replay over all 20 M points: indices, distances, intersection counts, rounds."""
    n, k, every = 20_000_000, 10, 97
    got = _run("hip", 4, n, k, "counter", tmp_path, 29691, {"HALO_LEVELS": "2", "GATHER_EVERY": str(every)}, timeout=1200)
    pts = datasets.uniform3d_counter(0, n, seed=0)
    q = np.arange(0, n, every, dtype=np.int32)
    ref = oracle.trueknn(pts, k, datasets.start_radius(n, k), query_ids=q)
    assert np.array_equal(got["gids"], q.astype(np.int64))
    assert np.array_equal(got["idx"], ref["idx"][q]) and np.array_equal(got["dist"], ref["dist"][q])
    assert np.array_equal(got["isect"], ref["intersections"][q])
    assert int(got["rounds"]) == ref["rounds"] and bool(got["one_pass"])
    assert 0 < int(got["halo_points"]) < int(got["tile"]) // 4


@pytest.mark.gpu
def test_config5_shaped_tiles_at_size(tmp_path):
    """... and BASELINE config 5's shape: 10 M heavy-tailed 2-D points over 4 tiles, RT-DBSCAN with the auto-grown eps (growth
    rounds that only count, one clustering, label propagation on clusters over the tiles).  Reference: the SINGLE-GPU engine's
    tknnDbscanAuto on the whole set in this process -- which tests/test_dbscan.py::test_config5_set_full_size_auto_eps checks
    against the CPU spec at 50 M points --: rounds, eps, noise count, cluster count, and labels and core flags of every 7th id."""
    from dist_worker import make_points
    from owlraytracing_amd.trueknn import TrueKNN
    n, eps0, min_pts, max_noise, every = 10_000_000, float(np.float32(0.00001)), 4, 0.02, 7
    got = _run("hip", 4, n, 4, "planar", tmp_path, 29692, {"DBSCAN_EPS": repr(eps0), "DBSCAN_MINPTS": str(min_pts), "DBSCAN_MAX_NOISE": repr(max_noise),
                                                            "GATHER_EVERY": str(every)}, timeout=1200)
    eng = TrueKNN()
    eng.build(make_points("planar", n))
    ref = eng.dbscan_auto(eps0, min_pts, max_noise)
    q = np.arange(0, n, every)
    assert ref["info"]["rounds"] > 1
    assert int(got["eps_rounds"]) == ref["info"]["rounds"] and float(got["eps"]) == float(np.float32(ref["info"]["eps"]))
    assert int(got["noise"]) == ref["info"]["noise"] and int(got["clusters"]) == ref["info"]["clusters"]
    assert np.array_equal(got["gids"], q)
    assert np.array_equal(got["labels"], ref["labels"].cpu().numpy()[q]) and np.array_equal(got["core"], ref["core"].cpu().numpy().astype(bool)[q])
    eng.close()
