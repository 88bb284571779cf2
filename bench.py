#!/usr/bin/env python3
"""Contract benchmark: TrueKNN queries/sec on BASELINE.json config 2 (10 M uniform 3-D points, k=10).

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A "step" is one whole radius-doubling TrueKNN solve (samples/s01-trueknn/hostCode.cpp:285-340) over
the rank's resident batch: points and the LBVH are in HBM before the timed region, like the
reference, which times "Build time" and "True KNN time" separately (hostCode.cpp:211,344).
With N GPUs the point set is N x 10 M points cut into Morton tiles (weak scaling): every step is
halo exchange over RCCL + halo-tree build + solve + termination all-reduce.

Prints ONE JSON line on rank 0.  Extra keys: roofline (dominant kernel, HBM bound, algorithmic
bytes per SURVEY 8d), cpu_baseline (the CPU checker's restatement timed on the host cores, rank 0,
N=1 only).
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

N_POINTS = 10_000_000  # BASELINE.json configs[1]
K = 10
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(n, k, total_intersections, total_active_rounds):
    """B_q = 12*R_q + 12*C_q + 8*k summed over queries."""
    return 12 * total_active_rounds + 12 * total_intersections + 8 * k * n


def cpu_baseline(xyz, k, r0, seconds_budget=25.0):
    """The CPU checker (oracle/, reference semantics) on a bounded sample of the same workload."""
    import oracle

    n = len(xyz)
    threads = oracle.num_threads()
    rng = np.random.default_rng(7)
    sample = 2_000_000  # ~10 s of CPU work on the GPU box: bounded, but long enough to average out scheduling noise
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
    # exact brute force (the north_star's "CPU brute-force"), bounded: 2000 queries x 10 M points
    qb = q[:2000]
    t0 = time.perf_counter()
    oracle.bruteforce_knn(xyz, k, qb)
    dtb = time.perf_counter() - t0
    out["bruteforce_exact_queries_per_s"] = len(qb) / dtb
    return out, ref, q


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=N_POINTS, help="points per GPU (default: BASELINE config 2)")
    ap.add_argument("--k", type=int, default=K)
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 lane, 2 wave")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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

    n, k = args.n, args.k
    sharded = world > 1 or args.sharded
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
        n_total = n * world
        r0 = datasets.start_radius(n_total, k)
        solver = tkd.ShardedTrueKNN(dev, kernel=args.kernel)
        solver.load_counter_based(n_total, seed=0)  # one-time: generate, Morton-tile, build own trees
        step = lambda: solver.solve(k, r0)  # noqa: E731
    else:
        dist = None
        n_total = n
        xyz_host = datasets.uniform3d(n, seed=0)
        r0 = datasets.start_radius(n, k)
        pts = torch.from_numpy(xyz_host).to(dev)
        eng = TrueKNN(device=local_rank)
        build_info = eng.build(pts)
        build_info = eng.build(pts)  # second build: steady-state build time (first one pays allocations)
        out = {}

        def step():
            r = eng.solve(k, r0, kernel=args.kernel, out=out)
            out.update({kk: v for kk, v in r.items() if kk != "info"})
            return r["info"]

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
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = elapsed / args.steps * 1e3
    value = n_total * args.steps / elapsed
    info = infos[-1]
    kern_ms = float(np.mean([i["dominant_kernel_ms"] for i in infos]))
    total_isect = int(info["total_intersections"])
    total_rounds_active = int(info["total_active_rounds"])
    n_local = len(solver.points) if sharded else n
    alg_bytes = algorithmic_bytes(n_local, k, total_isect, total_rounds_active)
    launches = max(int(info["dominant_kernel_launches"]), 1)
    achieved = alg_bytes / launches / (kern_ms * 1e-3) / 1e9
    kernel_name = {1: "lane_round_kernel", 2: "wave_packet_kernel", 3: "team_kernel"}.get(int(info["kernel_used"]), "?")
    # HBM traffic from the PMC counters is collected in separate rocprofv3 passes (profiles/);
    # attach the committed per-launch figure when it was measured for this exact workload.
    traffic, issue = None, {}
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            rec = json.load(open(tpath)).get("%s:n=%d:k=%d" % (kernel_name, n_local, k))
            if rec:
                traffic = rec["bytes_per_launch"]
                issue = {k2: rec[k2] for k2 in ("valu_wave_instructions_per_launch", "salu_wave_instructions_per_launch") if k2 in rec}
        except Exception:
            traffic = None

    line = {
        "metric": "kNN queries/sec (10M pts, k=10)",
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
            "workload": "TrueKNN on %d uniform-random 3-D points per GPU (numpy default_rng(0), [0,1)^3), k=%d, "
                        "start radius 0.25*(k/n)^(1/3)=%.6g; BASELINE.json configs[1]%s" % (
                            n_local, k, r0, "" if not sharded else "; %d Morton tiles, RCCL halo exchange" % world),
            "n_points_total": n_total,
            "k": k,
            "start_radius": r0,
            "kernel": kernel_name,
            "parallelism": "1 GPU" if not sharded else "%d Morton tiles + halo exchange" % world,
        },
        "rounds": int(info["rounds"]),
        "intersection_program_calls_per_s": total_isect * world / (ms_per_step * 1e-3) if world == 1 else None,
        "point_box_tests_per_s": int(info["point_tests"]) / (kern_ms * 1e-3),
        "node_box_tests_per_s": int(info["node_tests"]) / (kern_ms * 1e-3),
        "roofline": {
            "bound": "hbm",
            "kernel": kernel_name,
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "algorithmic_bytes_per_launch": alg_bytes // launches,
            "launches_per_step": launches,
            "kernel_ms": kern_ms,
            "timing": "HIP events on the launch stream, recorded inside libowl_mi355x.so around the kernel",
        },
    }
    if issue.get("valu_wave_instructions_per_launch"):
        # the bound that actually binds: vector instruction issue.  Instruction counts come
        # from the committed rocprofv3 PMC pass of this workload, the time is this run's; a wave64 VALU
        # instruction occupies its SIMD for 4 cycles, 4 SIMDs per CU.
        props = torch.cuda.get_device_properties(dev)
        clock_hz = float(getattr(props, "clock_rate", 2400000)) * 1e3
        simds = props.multi_processor_count * 4
        line["roofline"]["issue"] = {
            "valu_wave_instructions_per_launch": issue["valu_wave_instructions_per_launch"],
            "salu_wave_instructions_per_launch": issue.get("salu_wave_instructions_per_launch"),
            "valu_busy_frac": issue["valu_wave_instructions_per_launch"] * 4.0 / (simds * clock_hz * kern_ms * 1e-3),
            "clock_mhz": clock_hz / 1e6,
            "source": "profiles/hbm_traffic.json (rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU on this workload)",
        }
    if not sharded:
        line["build_ms"] = float(build_info["build_ms"])
        line["tree_bytes"] = int(build_info["device_bytes"])
    if rank == 0:
        # context for the roofline ("report against the measured stream peak as well"): a plain
        # device-to-device copy of 1 GiB on this GPU, read + write bytes over the best of 5 timings
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
        line["roofline"]["measured_copy_GBps"] = 2 * src.numel() * 4 / (best * 1e-3) / 1e9
        del src, dst
    if rank == 0 and not sharded and not args.no_cpu_baseline:
        cb, ref, q = cpu_baseline(xyz_host, k, r0)
        line["cpu_baseline"] = cb
        # the sample doubles as a parity spot check of the benchmarked run itself
        ql = torch.from_numpy(q.astype(np.int64)).to(dev)
        ok = (np.array_equal(out["idx"][ql].cpu().numpy(), ref["idx"][q])
              and np.array_equal(out["dist"][ql].cpu().numpy(), ref["dist"][q])
              and np.array_equal(out["intersections"][ql].cpu().numpy(), ref["intersections"][q]))
        line["parity_spot_check"] = "bit-exact on %d sampled rows" % len(q) if ok else "MISMATCH"
    if rank == 0:
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
