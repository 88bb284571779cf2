#!/usr/bin/env python3
"""Contract benchmark of the neighbour-query path (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W              TrueKNN, 10 M uniform points per GPU, k=10 (configs[1])
    python bench.py --workload dbscan                          RT-DBSCAN, 10 M Gaussian-mixture points, eps 0.01, minPts 4 (configs[2])
    python bench.py --gpus N --scaling strong [--points 100000000]  one fixed point set cut into N Morton tiles (configs[3] with 1e8)
    (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL)

A "step" is one whole solve over the rank's resident batch: points and the LBVH are in HBM before the
timed region -- the reference, too, times "Build time" and "True KNN time" separately
(samples/s01-trueknn/hostCode.cpp:211,344).  With N GPUs every step is halo selection + point-to-point
exchange + halo-tree build + solve + termination all-reduce.

Prints ONE JSON line on rank 0.  Beside the contract's keys: `roofline` (dominant kernel, HBM bound,
algorithmic bytes per SURVEY 8d, HIP-event kernel time; traffic and instruction counts from the committed
rocprofv3 PMC passes if they were taken on THESE sources), `cpu_baseline` (the CPU checker's restatement on
the host cores: rank 0, N = 1 only, bounded sample), `api_layout_writeback` (the solve with the reference's
24-byte frameBuffer records), and for N > 1 `ranks`, `phase_ms`, `halo_points`, `halo_exchanges`.
Exit status 1 (and "value": null) if the run's own parity spot check fails.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

N_POINTS = 10_000_000  # BASELINE.json configs[1] / configs[2]
K = 10
DB_EPS, DB_MINPTS = 0.01, 4
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(n, k, total_intersections, total_active_rounds):
    """B_q = 12*R_q + 12*C_q + 8*k summed over queries."""
    return 12 * total_active_rounds + 12 * total_intersections + 8 * k * n


def cpu_baseline_trueknn(xyz, k, r0):
    """The CPU checker (oracle/, reference semantics) on a bounded sample of the same workload."""
    import oracle

    n = len(xyz)
    threads = oracle.num_threads()
    rng = np.random.default_rng(7)
    sample = min(2_000_000, n)  # ~10 s of CPU work on the GPU box: bounded, but long enough to average out scheduling noise
    q = np.sort(rng.choice(n, sample, replace=False)).astype(np.int32)
    t0 = time.perf_counter()
    ref = oracle.trueknn(xyz, k, r0, query_ids=q)
    wall = time.perf_counter() - t0
    out = {
        "value": sample / ref["query_seconds"],
        "unit": "queries/s",
        "cores": threads,
        "kind": "port",
        "sample": "%d seeded queries of the %d-point workload through oracle/trueknn_oracle.c (OpenMP, %d "
                  "threads): %.2fs in the query loops; the per-round candidate grid over all points "
                  "(amortised over all n queries in a full run) is excluded, %.2fs wall in all" % (
                      sample, n, threads, ref["query_seconds"], wall),
        "rounds": int(ref["rounds"]),
    }
    # exact brute force (the north_star's "CPU brute-force"), bounded: 2000 queries x n points
    qb = q[:2000]
    t0 = time.perf_counter()
    oracle.bruteforce_knn(xyz, k, qb)
    dtb = time.perf_counter() - t0
    out["bruteforce_exact_queries_per_s"] = len(qb) / dtb
    return out, ref, q


def cpu_baseline_dbscan(xyz, eps, min_pts, budget_s=90.0):
    """the spec's grid DBSCAN on all host cores (oracle.dbscan_threaded).  The whole set
    if a timed 5 % slab says it fits the budget (config 3 takes 40 s on the 128 threads of a GPU box: the budget is
    generous because the whole set is also the bench's parity check), else the largest leading share of the points
    that does (clusters of a mixture keep their shape, the density falls with the share: the rate is then an upper bound)."""
    import oracle

    n = len(xyz)
    threads = oracle.num_threads()
    probe = max(n // 20, 1)
    t0 = time.perf_counter()
    oracle.dbscan_threaded(xyz[:probe], eps, min_pts)
    dt = time.perf_counter() - t0
    # work per point grows with the density, i.e. with the share taken: cost(share) ~ share^2 * cost(all)
    est_full = dt * (n / probe) ** 2
    share = 1.0 if est_full <= budget_s else max((budget_s / est_full) ** 0.5, probe / n)
    m = min(n, max(int(n * share), probe))
    t0 = time.perf_counter()
    r = oracle.dbscan_threaded(xyz[:m], eps, min_pts)
    wall = time.perf_counter() - t0
    return {
        "value": m / wall,
        "unit": "points/s",
        "cores": threads,
        "kind": "port",
        "sample": "the first %d of the %d points (%s) through oracle/dbscan_oracle.c:dbref_dbscan_mt (OpenMP, %d threads): "
                  "%.2fs wall = grid %.2f + core flags %.2f + unions %.2f + labels %.2f; %d clusters there" % (
                      m, n, "the whole workload" if m == n else "a share: lower density than the workload, so an upper bound on the CPU rate",
                      threads, wall, r["seconds"][0], r["seconds"][1], r["seconds"][2], r["seconds"][3], r["clusters"]),
    }, r, m


def committed_profile(kernel_name, n_local, k):
    """Per-launch PMC figures from profiles/hbm_traffic.json, if they were measured on these very sources."""
    from owlraytracing_amd import _lib

    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if not os.path.exists(path):
        return None, "no profiles/hbm_traffic.json"
    try:
        rec = json.load(open(path)).get("%s:n=%d:k=%d" % (kernel_name, n_local, k))
    except Exception as e:  # a damaged file is not a reason to lose the bench line
        return None, "profiles/hbm_traffic.json unreadable: %s" % e
    if not rec:
        return None, "no PMC record for this kernel and size"
    have, want = rec.get("source_sha16"), _lib.source_fingerprint(kernel_name)
    if have != want:
        return None, "the committed PMC record was taken on other kernel sources (%s, these are %s): re-run scripts/profile_gpu.sh" % (have, want)
    return rec, None


def copy_bandwidth(dev):
    """A plain device-to-device copy of 1 GiB on this GPU, read + write bytes over the best of 5 timings (context
    for the roofline, SURVEY 8d: "report against the measured stream peak as well")."""
    src = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    dst = torch.empty_like(src)
    best = None
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dst.copy_(src)
        e1.record()
        e1.synchronize()
        t = e0.elapsed_time(e1)
        best = t if best is None else min(best, t)
    del src, dst
    return 2 * (1 << 28) * 4 / (best * 1e-3) / 1e9


def sharded_trueknn_run(solver, dist, dev, backend, world, n_total, k, steps, warmup):
    """One timed run of the Morton-tile driver over the counter-based uniform set of `n_total` points (all ranks call it):
    load + tile + build (untimed), `warmup` solves, `steps` solves between barriers, the slowest rank's clock.  Returns what
    the JSON line (or one of its sub-objects) carries: value, ms_per_step, per-rank records, the slowest rank's phase times."""
    from owlraytracing_amd import datasets

    r0 = datasets.start_radius(n_total, k)
    solver.load_counter_based(n_total, seed=0)  # one-time: generate, Morton-tile, build own trees
    n_local = len(solver.points)
    for _ in range(warmup):
        solver.solve(k, r0)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    infos = []
    for _ in range(steps):
        infos.append(solver.solve(k, r0))
    torch.cuda.synchronize()
    dist.barrier()
    elapsed_own = time.perf_counter() - t0
    t = torch.tensor([elapsed_own], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    last = infos[-1]
    # one record per rank (an imbalance -- tile sizes come from 1 024 samples per rank -- must be visible in the one line a
    # hardware run leaves): points, halo rows, exchanges, this rank's own step time, and the per-phase wall times of two more,
    # instrumented steps (a device synchronisation after every phase, so they are not part of the timed region)
    names = ["select", "exchange", "halo_build", "solve", "reduce"]
    mine = [float(n_local), float(last.get("halo_points", 0)), float(last.get("halo_exchanges", 0)), elapsed_own / steps * 1e3]
    solver.profile = True
    ph = [solver.solve(k, r0)["phase_ms"] for _ in range(2)]
    solver.profile = False
    mine += [float(np.mean([p_.get(nm, 0.0) for p_ in ph])) for nm in names]
    mine.append(float(np.mean([i["dominant_kernel_ms"] for i in infos])))
    v = torch.tensor(mine, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")  # (gloo gathers host tensors only)
    every = [torch.empty_like(v) for _ in range(world)]
    dist.all_gather(every, v)
    rows = [e.tolist() for e in every]
    hp = torch.tensor([int(last.get("halo_points", 0))], dtype=torch.int64, device=dev)
    dist.all_reduce(hp, op=dist.ReduceOp.SUM)
    return {
        "value": n_total * steps / elapsed,
        "ms_per_step": elapsed / steps * 1e3,
        "steps": steps,
        "warmup": warmup,
        "n_points_total": n_total,
        "start_radius": r0,
        "n_local": n_local,
        "infos": infos,
        "ranks": [{"rank": j, "points": int(r_[0]), "halo_points": int(r_[1]), "halo_exchanges": int(r_[2]), "ms_per_step_own_clock": r_[3],
                   "phase_ms": {nm: r_[4 + i] for i, nm in enumerate(names)}, "kernel_ms": r_[4 + len(names)]} for j, r_ in enumerate(rows)],
        "phase_ms": {nm: max(r_[4 + i] for r_ in rows) for i, nm in enumerate(names)},  # the slowest rank of each phase
        "halo_points": int(hp.item()),
        "halo_exchanges": int(last.get("halo_exchanges", 0)),
    }


def sharded_dbscan_records(solver, dist, dev, backend, world, eps32, min_pts, infos, n_local, ms_own):
    """What a sharded RT-DBSCAN line carries beside its value (it had neither phases nor kernel times nor a
    roofline): one record per rank -- points, halo, label rounds, its own step time, the phases of two more, instrumented
    steps (set-up: halo selection + exchange + tree; the tile's clustering with its three traversal kernels' device times;
    label propagation; numbering; border assignment) -- the slowest rank of each phase, and rank 0's per-kernel rooflines
    (the tile's own + halo points: the set its engine clustered)."""
    last = infos[-1]
    solver.profile = True
    prof = [solver.dbscan(eps32, min_pts)["info"] for _ in range(2)]
    solver.profile = False
    names = ["setup", "cluster", "propagate", "number", "assign"]
    knames = ["core_ms", "union_ms", "label_ms", "solve_ms"]
    eng = [p_.get("engine") or {} for p_ in prof]
    mine = [float(n_local), float(last.get("halo_points", 0)), float(last.get("label_rounds", last.get("rounds", 0))), ms_own]
    mine += [float(np.mean([p_.get("phase_ms", {}).get(nm, 0.0) for p_ in prof])) for nm in names]
    mine += [float(np.mean([e_.get(nm, 0.0) for e_ in eng])) for nm in knames]
    v = torch.tensor(mine, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    every = [torch.empty_like(v) for _ in range(world)]
    dist.all_gather(every, v)
    rows = [e.tolist() for e in every]
    hp = torch.tensor([int(last.get("halo_points", 0))], dtype=torch.int64, device=dev)
    dist.all_reduce(hp, op=dist.ReduceOp.SUM)
    out = {
        "ranks": [{"rank": j, "points": int(r_[0]), "halo_points": int(r_[1]), "label_rounds": int(r_[2]), "ms_per_step_own_clock": r_[3],
                   "phase_ms": {nm: r_[4 + i] for i, nm in enumerate(names)},
                   "kernel_ms": {nm: r_[4 + len(names) + i] for i, nm in enumerate(knames)}} for j, r_ in enumerate(rows)],
        "phase_ms": {nm: max(r_[4 + i] for r_ in rows) for i, nm in enumerate(names)},
        "halo_points": int(hp.item()),
        "halo_exchanges": int(last.get("halo_exchanges", 1)),
        "label_rounds": int(last.get("label_rounds", last.get("rounds", 0))),
    }
    if eng and eng[-1].get("groups") is not None and "union_node_tests" in eng[-1]:
        kernels = dbscan_rooflines(eng[-1], eng, int(eng[-1].get("n_clustered", n_local)), min_pts)
        dom = max(kernels, key=lambda nm: kernels[nm]["kernel_ms"] * kernels[nm]["launches_per_step"])
        out["roofline"] = dict(kernels[dom])
        out["roofline"]["kernels"] = kernels
        out["roofline"]["of"] = "rank 0's tile and its 2-eps halo (%d points)" % int(eng[-1].get("n_clustered", n_local))
    return out


def exchange_label(backend):
    """What carried the halo rows in THIS run (the line said "RCCL" whatever the backend)."""
    return "RCCL (nccl backend) halo exchange" if backend == "nccl" else "%s-backend rehearsal: halo rows staged through the host, ranks may share a GPU" % backend


def sub_object(run, world, k, what):
    """A second workload of the same multi-rank run, as a sub-object of the JSON line (same clock rules as the headline)."""
    return {"metric": "kNN queries/sec (%s pts, k=%d)" % (size_label(run["n_points_total"]), k), "value": run["value"], "unit": "queries/s",
            "n_gpus": world, "steps": run["steps"], "warmup": run["warmup"], "ms_per_step": run["ms_per_step"], "scaling": what,
            "n_points_total": run["n_points_total"], "start_radius": run["start_radius"], "ranks": run["ranks"], "phase_ms": run["phase_ms"],
            "halo_points": run["halo_points"], "halo_exchanges": run["halo_exchanges"],
            "kernel_ms_mean_over_ranks": float(np.mean([r_["kernel_ms"] for r_ in run["ranks"]]))}


def size_label(n):
    return "%dM" % (n // 1_000_000) if n % 1_000_000 == 0 else str(n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=("trueknn", "dbscan"), default="trueknn")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None,
                    help="weak: --points per GPU; strong: --points in all, cut into one Morton tile per GPU.  Default with N > 1 and no "
                         "--points: BASELINE's 10 M set cut into N tiles (strong) as the headline, N x 10 M (weak) and, at N = 8, the 100 M set "
                         "of configs[3] as sub-objects of the line")
    # (not "--n": torch.distributed.run's own parser stumbles over script options that abbreviate its --nnodes / --nproc-per-node)
    ap.add_argument("--points", dest="n", type=int, default=None, help="points per GPU (weak) or in all (strong); default: BASELINE config 2's 10 M")
    ap.add_argument("--k", type=int, default=K)
    ap.add_argument("--eps", type=float, default=DB_EPS)
    ap.add_argument("--min-pts", type=int, default=DB_MINPTS)
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 lane, 2 wave, 3 team")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fb-leg", action="store_true",
                    help="skip the api_layout_writeback solves (scripts/profile_gpu.sh: their launches write a 2.4 GB frameBuffer "
                         "and must not be averaged into the PMC record of the benchmarked launch)")
    ap.add_argument("--no-dbscan-leg", action="store_true", help="skip the RT-DBSCAN config-3 object of the default run")
    ap.add_argument("--sharded", action="store_true", help="use the Morton-tile / halo-exchange driver even for one rank")
    args = ap.parse_args()

    from owlraytracing_amd import _lib, datasets
    from owlraytracing_amd.trueknn import TrueKNN

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    # TKNN_BENCH_BACKEND=gloo: rehearsal of the N > 1 path with the ranks sharing the GPUs there are
    # (messages staged through the host); the driver's runs use nccl = RCCL, one rank per GPU
    backend = os.environ.get("TKNN_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    k = args.k
    sharded = world > 1 or args.sharded
    dbscan = args.workload == "dbscan"
    # N > 1 without --scaling / --points: the headline is BASELINE's OWN set -- configs[1], 10 M points -- cut into N Morton tiles
    # ("kNN queries/sec (10M pts, k=10) at 1/2/4/8 MI355X": strong scaling), with N x 10 M (weak) and, at N = 8, configs[3]
    # (100 M points, strong) as sub-objects of the same line (the default used to time N x 10 M under the 10 M metric)
    default_multi = sharded and not dbscan and args.scaling is None and args.n is None
    scaling = args.scaling or ("strong" if default_multi else "weak")
    n_arg = args.n if args.n is not None else N_POINTS
    n_total = n_arg if (scaling == "strong" or not sharded) else n_arg * world
    eps32 = float(np.float32(args.eps))
    dist = None
    extra = {}
    run = None
    if sharded:
        import torch.distributed as dist

        from owlraytracing_amd import distributed as tkd

        if not dist.is_initialized():
            if "RANK" not in os.environ:  # plain `python bench.py --sharded`: a one-rank group
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29577")
                os.environ.setdefault("RANK", "0")
                os.environ.setdefault("WORLD_SIZE", "1")
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(backend)
        solver = tkd.ShardedTrueKNN(dev, kernel=args.kernel)
    if sharded and not dbscan:
        run = sharded_trueknn_run(solver, dist, dev, backend, world, n_total, k, args.steps, args.warmup)
        r0, n_local, infos = run["start_radius"], run["n_local"], run["infos"]
        elapsed = run["ms_per_step"] * args.steps / 1e3
        for key in ("ranks", "phase_ms", "halo_points", "halo_exchanges"):
            extra[key] = run[key]
        if default_multi and world > 1:
            few = max(1, min(args.steps, 5))
            weak = sharded_trueknn_run(solver, dist, dev, backend, world, N_POINTS * world, k, few, 1)
            extra["weak"] = sub_object(weak, world, k, "weak")
            if world == 8:
                c4 = sharded_trueknn_run(solver, dist, dev, backend, world, 100_000_000, k, few, 1)
                extra["config4"] = sub_object(c4, world, k, "strong")
                extra["config4"]["config"] = "BASELINE.json configs[3]: TrueKNN on 100M uniform 3-D points, k=10, Morton-tiled across 8 GPUs"
    elif sharded:
        # every rank draws the whole mixture's slice it is given (the generator is sequential: slices of one stream)
        lo, hi = n_total * rank // world, n_total * (rank + 1) // world
        pts = datasets.gaussian_mixture3d(n_total, components=64, sigma=0.02, seed=1)[lo:hi]
        solver.load_points(torch.from_numpy(pts), torch.arange(lo, hi, dtype=torch.int32))
        step = lambda: solver.dbscan(eps32, args.min_pts)["info"]  # noqa: E731
        r0 = None
        n_local = len(solver.points)
    else:
        if dbscan:
            xyz_host = datasets.gaussian_mixture3d(n_total, components=64, sigma=0.02, seed=1)
            r0 = None
        else:
            xyz_host = datasets.uniform3d(n_total, seed=0)
            r0 = datasets.start_radius(n_total, k)
        pts = torch.from_numpy(xyz_host).to(dev)
        eng = TrueKNN(device=local_rank)
        build_info = eng.build(pts)
        build_info = eng.build(pts)  # second build: steady-state build time (first one pays allocations)
        out = {}
        n_local = n_total

        if dbscan:
            def step():
                r = eng.dbscan(eps32, args.min_pts)
                out.update({kk: v for kk, v in r.items() if kk != "info"})
                return r["info"]
        else:
            def step():
                r = eng.solve(k, r0, kernel=args.kernel, out=out)
                out.update({kk: v for kk, v in r.items() if kk != "info"})
                return r["info"]

    if run is None:
        for _ in range(args.warmup):
            info = step()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        infos = []
        for _ in range(args.steps):
            infos.append(step())
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        elapsed_own = elapsed
        if dist is not None:  # (sharded RT-DBSCAN)
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            extra.update(sharded_dbscan_records(solver, dist, dev, backend, world, eps32, args.min_pts, infos, n_local, elapsed_own / args.steps * 1e3))

    ms_per_step = elapsed / args.steps * 1e3
    value = n_total * args.steps / elapsed
    info = infos[-1]
    tiles = "" if not sharded else "; %d Morton tiles, %s" % (world, exchange_label(backend))
    if dbscan:
        line = dbscan_line(args, info, infos, value, ms_per_step, n_total, n_local, world, sharded, eps32, tiles)
        if sharded and "roofline" in extra:
            line["roofline"] = extra.pop("roofline")
    else:
        line = trueknn_line(args, info, infos, value, ms_per_step, n_total, n_local, world, sharded, r0, tiles, scaling)
    line["scaling"] = scaling if sharded else "weak"
    line["backend"] = backend if sharded else None
    line.update(extra)
    checks = {}  # named result checks of this run; ANY red one fails the run (run_failed)
    if not sharded:
        line["build_ms"] = float(build_info["build_ms"])
        line["tree_bytes"] = int(build_info["device_bytes"])
    if rank == 0:
        line["roofline"]["measured_copy_GBps"] = copy_bandwidth(dev)
    if rank == 0 and not sharded and not dbscan and not args.no_fb_leg:
        # the reference's result layout (24-byte Neigh records, GeomTypes.h:22-28) is an artefact of
        # its API; its write-back cost is reported apart from the compact rows the step writes
        fb_ms = []
        for _ in range(3):
            r = eng.solve(k, r0, kernel=args.kernel, want_fb=True)
            fb_ms.append(float(r["info"]["solve_ms"]))
        del r
        plain = float(np.mean([i["solve_ms"] for i in infos]))
        line["api_layout_writeback"] = {
            "solve_ms_with_frameBuffer": float(np.min(fb_ms)),
            "solve_ms_compact_rows_only": plain,
            "extra_ms": float(np.min(fb_ms)) - plain,
            "frameBuffer_bytes": 24 * k * n_total,
            "note": "one solve writing idx/dist/intersections AND the n*k 24-byte records the reference's host loop reads",
        }
    if rank == 0 and not sharded and not dbscan and not args.no_dbscan_leg and n_total == N_POINTS:
        # BASELINE config 3 in the same run (the driver only runs the default command): timed like the steps
        # above, per-kernel rooflines, a sampled eps-ball parity check by the CPU spec
        db_failed, line["dbscan_config3"] = dbscan_config3_leg(args, eng, dev)
        checks["dbscan_config3"] = not db_failed
    if rank == 0 and not sharded and not args.no_cpu_baseline:
        if dbscan:
            cb, ref, m = cpu_baseline_dbscan(xyz_host, eps32, args.min_pts)
            line["cpu_baseline"] = cb
            if m == n_total:  # the whole set went through the checker: compare the benchmarked run with it
                ok = (np.array_equal(out["labels"].cpu().numpy(), ref["labels"]) and np.array_equal(out["core"].cpu().numpy(), ref["core"]))
                line["parity_spot_check"] = "labels and core flags of all %d points equal the CPU spec's" % m if ok else "MISMATCH"
                checks["parity_spot_check"] = ok
            else:
                # core flags of the first m points are density-dependent: no comparison on a share; the full-size parity
                # test is tests/test_dbscan.py::test_config3_full_size (-m gpu)
                line["parity_spot_check"] = "not applicable on a share of the set (see tests/test_dbscan.py, full size)"
        else:
            cb, ref, q = cpu_baseline_trueknn(xyz_host, k, r0)
            line["cpu_baseline"] = cb
            # the sample doubles as a parity spot check of the benchmarked run itself
            ql = torch.from_numpy(q.astype(np.int64)).to(dev)
            ok = (np.array_equal(out["idx"][ql].cpu().numpy(), ref["idx"][q])
                  and np.array_equal(out["dist"][ql].cpu().numpy(), ref["dist"][q])
                  and np.array_equal(out["intersections"][ql].cpu().numpy(), ref["intersections"][q]))
            line["parity_spot_check"] = "bit-exact on %d sampled rows" % len(q) if ok else "MISMATCH"
            checks["parity_spot_check"] = ok
    failed = run_failed(checks)
    if failed:
        line["value"] = None  # a wrong result has no throughput
    if rank == 0:
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()
    if failed:
        sys.exit(1)


def run_failed(checks):
    """A run fails if ANY of its named checks is red (the kNN spot check used to overwrite the verdict of the
    dbscan_config3 leg, so a wrong DBSCAN result left exit status 0 and the headline value in place)."""
    return any(not ok for ok in checks.values())


def trueknn_line(args, info, infos, value, ms_per_step, n_total, n_local, world, sharded, r0, tiles, scaling="weak"):
    k = args.k
    kern_ms = float(np.mean([i["dominant_kernel_ms"] for i in infos]))
    total_isect = int(info["total_intersections"])
    total_rounds_active = int(info["total_active_rounds"])
    alg_bytes = algorithmic_bytes(n_local, k, total_isect, total_rounds_active)
    launches = max(int(info["dominant_kernel_launches"]), 1)
    achieved = alg_bytes / launches / (kern_ms * 1e-3) / 1e9
    kernel_name = {1: "lane_round_kernel", 2: "wave_packet_kernel", 3: "team_kernel"}.get(int(info["kernel_used"]), "?")
    rec, why_not = committed_profile(kernel_name, n_local, k)
    line = {
        # (the metric names the set that was timed: BASELINE's "10M pts, k=10" only when that is what ran)
        "metric": "kNN queries/sec (%s pts, k=%d)" % (size_label(n_total), k),
        "value": value,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "TrueKNN on %d uniform-random 3-D points%s (%s), k=%d, start radius 0.25*(k/n)^(1/3)=%.6g; "
                        "BASELINE.json configs[%d]%s" % (
                            n_total,
                            " in all, cut into %d tiles" % world if sharded and scaling == "strong" else (" in all, %d per GPU" % (n_total // world) if sharded else ""),
                            "counter-based Philox(0), [0,1)^3" if sharded else "numpy default_rng(0), [0,1)^3", k, r0,
                            3 if n_total >= 100_000_000 and sharded else 1, tiles),
            "ranks": world,
            "n_points_total": n_total,
            "k": k,
            "start_radius": r0,
            "kernel": kernel_name,
            "parallelism": "1 GPU" if not sharded else "%d Morton tiles + halo exchange" % world,
        },
        "rounds": int(info["rounds"]),
        "intersection_program_calls_per_s": total_isect * world / (ms_per_step * 1e-3) if world == 1 else None,
        "point_box_tests_per_s": int(info["point_tests"]) / (kern_ms * 1e-3),
        # exact point-in-box tests executed per intersection-program call of the reference (VERDICT r2 asked for < 7: leaf blocks
        # of 16 points against final boxes of some 80 candidates, COUNT and SELECT over the same blocks for a share of the queries)
        "point_tests_per_intersection": int(info["point_tests"]) / max(total_isect, 1),
        "node_box_tests_per_s": int(info["node_tests"]) / (kern_ms * 1e-3),
        "roofline": {
            "bound": "hbm",
            "kernel": kernel_name,
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": rec["bytes_per_launch"] if rec else None,
            "algorithmic_bytes_per_launch": alg_bytes // launches,
            "launches_per_step": launches,
            "kernel_ms": kern_ms,
            "timing": "HIP events on the launch stream, recorded inside libowl_mi355x.so around the kernel",
        },
    }
    if why_not:
        line["roofline"]["traffic_note"] = why_not
    if rec and rec.get("valu_wave_instructions_per_launch"):
        # what the kernel is really limited by: instruction issue and the latencies between
        # instructions, not HBM.  Counts and wait shares come from the committed PMC passes (same sources, checked
        # above), the time is this run's.  A wave64 VALU instruction occupies its SIMD for 2 cycles
        # (MI355X_MICROARCH.md, "Wave scheduling"); the instructions of this kernel that read or write lane masks,
        # cross lanes or compare 64-bit keys take about 4 (scripts/microbench/issue_rate.hip), so the first
        # figure is a lower bound of the pipe's occupancy.
        props = torch.cuda.get_device_properties(torch.cuda.current_device())
        clock_hz = float(getattr(props, "clock_rate", 2400000)) * 1e3
        simds = props.multi_processor_count * 4
        cyc = simds * clock_hz * kern_ms * 1e-3
        issue = {
            "valu_wave_instructions_per_launch": rec["valu_wave_instructions_per_launch"],
            "salu_wave_instructions_per_launch": rec.get("salu_wave_instructions_per_launch"),
            "valu_issue_frac_at_2_cycles": rec["valu_wave_instructions_per_launch"] * 2.0 / cyc,
            "clock_mhz": clock_hz / 1e6,
            "source": "profiles/hbm_traffic.json (rocprofv3 --pmc passes over this workload, sources %s)" % rec.get("source_sha16"),
        }
        for kk in ("wave_wait_frac", "wave_issue_stall_frac", "wave_active_frac", "scalar_pipe_instructions_per_launch",
                   "lds_instructions_per_launch"):
            if kk in rec:
                issue[kk] = rec[kk]
        if rec.get("scalar_pipe_instructions_per_launch"):
            # one scalar pipe per CU, about one instruction per cycle (issue_rate.hip: 4.4 cycles per instruction per SIMD)
            issue["scalar_pipe_frac"] = rec["scalar_pipe_instructions_per_launch"] / (props.multi_processor_count * clock_hz * kern_ms * 1e-3)
        line["roofline"]["issue"] = issue
    return line


def dbscan_line(args, info, infos, value, ms_per_step, n_total, n_local, world, sharded, eps32, tiles):
    line = {
        "metric": "RT-DBSCAN points/sec (%s Gaussian-mixture pts, eps=%.6g, minPts=%d)" % (size_label(n_total), eps32, args.min_pts),
        "value": value,
        "unit": "points/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "RT-DBSCAN on %d 3-D points of a 64-component Gaussian mixture (sigma 0.02, numpy default_rng(1)), eps=%.6g, "
                        "minPts=%d; BASELINE.json configs[2]%s" % (n_total, eps32, args.min_pts, tiles),
            "n_points_total": n_total,
            "eps": eps32,
            "min_pts": args.min_pts,
            "parallelism": "1 GPU" if not sharded else "%d Morton tiles + 2-eps halo + label propagation" % world,
        },
        "clusters": int(info.get("clusters", -1)),
    }
    if sharded:
        # (main() replaces this with rank 0's per-kernel rooflines: sharded_dbscan_records)
        line["roofline"] = {"bound": "hbm", "kernel": "db_group_union_kernel", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": None, "traffic": None, "note": "no engine record of the tile's clustering in this run"}
        return line
    kernels = dbscan_rooflines(info, infos, n_local, args.min_pts)
    # the dominant traversal kernel of the call, by its own HIP-event time (all of them under "kernels")
    dom = max(kernels, key=lambda nm: kernels[nm]["kernel_ms"] * kernels[nm]["launches_per_step"])
    line["roofline"] = dict(kernels[dom])
    line["roofline"].update({
        "groups": int(info.get("groups", 0)),
        "all_kernels_ms": {"core_flags": kernels["db_core_kernel"]["kernel_ms"],
                           "unions": kernels["db_group_union_kernel"]["kernel_ms"] * kernels["db_group_union_kernel"]["launches_per_step"],
                           "labels": kernels["db_label_kernel"]["kernel_ms"], "whole_call": float(np.mean([i["solve_ms"] for i in infos]))},
        "point_distance_tests": {"core_flags": int(info["core_point_tests"]), "unions": int(info["union_point_tests"]),
                                 "labels": int(info["label_point_tests"])},
        "node_box_tests": int(info["node_tests"]),
        "kernels": kernels,
    })
    return line


def dbscan_rooflines(info, infos, n_local, min_pts):
    """HBM roofline of each traversal kernel of a tknnDbscan call (core flags, group unions, labels), by the kernel's own
    HIP-event time.  SURVEY 8(d) carried over to DBSCAN: 12 B per point whose distance to a query is computed (the
    query's own 12 B once per traversal) + what the kernel writes per point (core flag 1 B / label 4 B); the union
    kernel walks once per packet of 64 groups, not per point: 32 B per node box it looks at + 32 B per group it serves.
    `traffic` = HBM bytes per launch from the committed PMC record of that kernel (null, with the reason, unless it was
    taken on these sources); `traffic_over_algorithmic` well above 1 = wasted re-reads / partial-sector writes."""
    names = {"core_ms": ("db_core_kernel", "core_point_tests", 1), "union_ms": ("db_group_union_kernel", "union_point_tests", 0),
             "label_ms": ("db_label_kernel", "label_point_tests", 4)}
    mean = {nm: float(np.mean([i[nm] for i in infos])) for nm in names}
    out = {}
    for nm, (kernel_name, tests_key, out_bytes) in names.items():
        launches = max(int(info.get("union_launches", 1)), 1) if nm == "union_ms" else 1
        if nm == "union_ms":
            alg_bytes = (32 * int(info["union_node_tests"]) + 12 * int(info[tests_key]) + 32 * int(info["groups"]) * launches) // launches
        else:
            alg_bytes = 12 * int(info[tests_key]) + 12 * n_local + out_bytes * n_local
        kern_ms = mean[nm] / launches
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        r = {
            "bound": "hbm",
            "kernel": kernel_name,
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": None,
            "algorithmic_bytes_per_launch": alg_bytes,
            "launches_per_step": launches,
            "kernel_ms": kern_ms,
            "timing": "HIP events on the launch stream, recorded inside libowl_mi355x.so around the kernel's launches"
                      + (" (the label kernel's launch; the walks of the points that are not core -- whose point tests these are -- run in "
                         "db_border_walk_kernel beside the unions, on a stream of their own)" if nm == "label_ms" else "")
                      + (" (the two launches; db_uniform_kernel between them is not in it)" if nm == "union_ms" else ""),
        }
        rec, why_not = committed_profile(kernel_name, n_local, min_pts)
        if rec and nm == "label_ms":
            # round 4: the labels stay by slot in db_label_kernel and a second launch, db_rows_from_slots_kernel, carries them to
            # the rows (label_ms spans both): the pass's traffic is the two kernels' together
            rec2, why2 = committed_profile("db_rows_from_slots_kernel", n_local, min_pts)
            if rec2:
                rec = dict(rec)
                for kk in ("bytes_per_launch", "FETCH_SIZE_KB_per_launch", "WRITE_SIZE_KB_per_launch"):
                    rec[kk] = rec[kk] + rec2[kk]
                r["kernels_of_the_pass"] = ["db_label_kernel", "db_rows_from_slots_kernel"]
                r["traffic_note"] = ("most of it is the gather's FETCH_SIZE: one 64-byte fabric request per scattered 4-byte word of a 40 MB array the "
                                     "launch before has written (served by the 256 MiB Infinity Cache, whose hits the counter includes); the x2 of wide "
                                     "coalesced reads (MI355X_MICROARCH.md, HBM) is applied to it as to every kernel although that width is uncalibrated: an "
                                     "upper bound.  Written: 40 MB by slot + 49 MB by row, where the scatter of round 3 wrote 648 MB")
            else:
                rec, why_not = None, "db_rows_from_slots_kernel: " + str(why2)
        if rec:
            r["traffic"] = rec["bytes_per_launch"]
            r["traffic_over_algorithmic"] = rec["bytes_per_launch"] / max(alg_bytes, 1)
            for kk in ("wave_wait_frac", "FETCH_SIZE_KB_per_launch", "WRITE_SIZE_KB_per_launch"):
                if kk in rec:
                    r[kk] = rec[kk]
        elif why_not:
            r["traffic_note"] = why_not
        out[kernel_name] = r
    return out


def dbscan_config3_leg(args, eng, dev, steps=5, warmup=1):
    """BASELINE configs[2] inside the default run: 10 M Gaussian-mixture points, eps 0.01, minPts 4 through tknnDbscan on
    the same engine object (new build).  Returns (failed, object for the JSON line)."""
    from owlraytracing_amd import datasets

    n = N_POINTS
    eps32 = float(np.float32(DB_EPS))
    xyz = datasets.gaussian_mixture3d(n, components=64, sigma=0.02, seed=1)
    build_info = eng.build(torch.from_numpy(xyz).to(dev))
    for _ in range(warmup):
        r = eng.dbscan(eps32, DB_MINPTS)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    infos = []
    for _ in range(steps):
        r = eng.dbscan(eps32, DB_MINPTS)
        infos.append(r["info"])
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    info = infos[-1]
    obj = {
        "metric": "RT-DBSCAN points/sec (10M Gaussian-mixture pts, eps=0.01, minPts=4)",
        "value": n * steps / elapsed,
        "unit": "points/s",
        "ms_per_step": elapsed / steps * 1e3,
        "steps": steps,
        "warmup": warmup,
        "config": {"workload": "RT-DBSCAN on %d 3-D points of a 64-component Gaussian mixture (sigma 0.02, numpy default_rng(1)), "
                               "eps=%.6g, minPts=%d; BASELINE.json configs[2]" % (n, eps32, DB_MINPTS)},
        "clusters": int(info["clusters"]),
        "groups": int(info.get("groups", 0)),
        "device_ms_whole_call": float(np.mean([i["solve_ms"] for i in infos])),
        "build_ms": float(build_info["build_ms"]),
        "roofline": dbscan_rooflines(info, infos, n, DB_MINPTS),
        "point_distance_tests": {"core_flags": int(info["core_point_tests"]), "unions": int(info["union_point_tests"]),
                                 "labels": int(info["label_point_tests"])},
    }
    failed = False
    if not args.no_cpu_baseline:
        import oracle

        t0 = time.perf_counter()
        sample = np.random.default_rng(33).choice(n, 2000, replace=False).astype(np.int32)
        lab, core = r["labels"].cpu().numpy(), r["core"].cpu().numpy()
        chk = oracle.dbscan_ball_check(xyz, eps32, DB_MINPTS, lab, core, sample)
        ok = chk["violations"] == 0 and int(lab.max()) + 1 == int(info["clusters"]) and bool(np.all(lab[core.astype(bool)] >= 0))
        obj["parity_spot_check"] = (
            "core flags and labels of %d sampled points equal what the CPU spec derives from their recomputed eps-balls "
            "(oracle/dbscan_oracle.c:dbref_ball_check, %.1fs); all %d points against the threaded CPU spec: "
            "tests/test_dbscan.py::test_config3_full_size and `bench.py --workload dbscan`" % (len(sample), time.perf_counter() - t0, n)
            if ok else "MISMATCH (%d of %d sampled points, first %d)" % (chk["violations"], len(sample), chk["first_bad"]))
        failed = not ok
        if failed:
            obj["value"] = None
    else:
        obj["parity_spot_check"] = "skipped (--no-cpu-baseline)"
    return failed, obj


if __name__ == "__main__":
    main()
