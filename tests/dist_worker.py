"""Worker for the multi-rank tests: launched by torch.distributed.run (see test_distributed.py).

    dist_worker.py <engine: checker|hip> <n_total> <k> <dataset> <out.npz>
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from owlraytracing_amd import datasets, distributed as tkd  # noqa: E402


class CheckerEngine:
    """Stand-in for owlraytracing_amd.trueknn.TrueKNN backed by the CPU checker's numpy restatement
    (tests only): same build / set_halo / solve contract, tensors on the CPU."""

    def __init__(self, device):
        self.device = device
        self.halo = (np.zeros((0, 3), np.float32), np.zeros(0, np.int32))

    def build(self, points, ids=None):
        self.pts = points.cpu().numpy()
        self.ids = ids.cpu().numpy()
        self.n = len(self.pts)

    def set_halo(self, points=None, ids=None):
        if points is None or len(points) == 0:
            self.halo = (np.zeros((0, 3), np.float32), np.zeros(0, np.int32))
        else:
            self.halo = (points.cpu().numpy(), ids.cpu().numpy())

    def solve(self, k, start_radius, kernel=0, max_rounds=64, want_fb=False, want_levels=False, allow_unfinished=False, out=None, phase=0):
        from oracle.trueknn_numpy import trueknn_numpy
        xyz = np.concatenate([self.pts, self.halo[0]])
        ids = np.concatenate([self.ids, self.halo[1]]).astype(np.int64)
        r = trueknn_numpy(xyz, k, start_radius, max_rounds=max_rounds, query_ids=np.arange(self.n),
                          stop_quietly=allow_unfinished, ids=ids)
        lev = r["level"][: self.n]
        if phase == 3:
            # tknnSolveOptions.phase = 3: only the rows an earlier call left unfinished (level -1) are solved; the others stay
            todo = out["levels"].numpy() < 0
            idx, dst, isect, levels = out["idx"].numpy(), out["dist"].numpy(), out["intersections"].numpy(), out["levels"].numpy()
            done_now = todo & (lev >= 0)
            idx[done_now], dst[done_now] = r["idx"][: self.n][done_now].astype(np.int32), r["dist"][: self.n][done_now]
            isect[done_now], levels[done_now] = r["intersections"][: self.n][done_now], lev[done_now].astype(np.int32)
            info = {"rounds": int(lev[done_now].max()) + 1 if done_now.any() else 0, "unfinished": int((todo & (lev < 0)).sum()),
                    "total_intersections": int(r["intersections"][: self.n][done_now].sum()), "kernel_used": 0,
                    "dominant_kernel_ms": 0.0, "dominant_kernel_launches": 1, "node_tests": 0, "point_tests": 0,
                    "total_active_rounds": 0, "solve_ms": 0.0, "final_radius": r["final_radius"], "list_capacity": k,
                    "tie_rows": 0, "tie_rows_left": 0, "tie_ms": 0.0}
            return {"idx": out["idx"], "dist": out["dist"], "intersections": out["intersections"], "levels": out["levels"], "info": info}
        info = {"rounds": int(lev.max()) + 1 if (lev >= 0).any() else r["rounds"], "unfinished": int((lev < 0).sum()),
                "total_intersections": int(r["intersections"][: self.n][lev >= 0].sum()), "kernel_used": 0,
                "dominant_kernel_ms": 0.0, "dominant_kernel_launches": 1, "node_tests": 0, "point_tests": 0,
                "total_active_rounds": 0, "solve_ms": 0.0, "final_radius": r["final_radius"], "list_capacity": k,
                "tie_rows": 0, "tie_rows_left": 0, "tie_ms": 0.0}
        return {"idx": torch.from_numpy(r["idx"][: self.n].astype(np.int32)),
                "dist": torch.from_numpy(r["dist"][: self.n]),
                "intersections": torch.from_numpy(r["intersections"][: self.n]),
                "levels": torch.from_numpy(lev.astype(np.int32)), "info": info}


    # ---- DBSCAN: the CPU spec (oracle/dbscan_oracle.c) behind the same two calls as the HIP engine ----
    def dbscan(self, eps, min_pts, want_counts=False):
        import oracle
        r = oracle.dbscan(self.pts, eps, min_pts)
        return {"labels": torch.from_numpy(r["labels"].astype(np.int32)), "core": torch.from_numpy(r["core"].astype(bool)),
                "info": {"clusters": int(r["clusters"])}}

    def dbscan_noise(self, eps, min_pts):
        import oracle
        r = oracle.dbscan(self.pts, eps, min_pts)
        flags = r["labels"] < 0
        return {"noise": torch.from_numpy(flags), "count": int(flags.sum())}

    def dbscan_assign(self, eps, core_label):
        from scipy.spatial import cKDTree
        lab = np.asarray(core_label.cpu().numpy(), np.int32)
        out = lab.copy()
        tree = cKDTree(self.pts.astype(np.float64))
        eps = np.float32(eps)
        for q in np.nonzero(lab < 0)[0]:
            best = -1
            for p in tree.query_ball_point(self.pts[q].astype(np.float64), float(eps) * 1.0001 + 1e-30):
                if lab[p] < 0:
                    continue
                d = self.pts[p] - self.pts[q]
                if np.sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2], dtype=np.float32) <= eps and (best < 0 or lab[p] < best):
                    best = lab[p]
            out[q] = best
        return torch.from_numpy(out)


def make_points(name, n):
    if name == "counter":  # bench.py's set: defined independently of how it is cut (datasets.uniform3d_counter)
        return datasets.uniform3d_counter(0, n, seed=0)
    if name == "uniform":
        return datasets.uniform3d(n, seed=5)
    if name == "clustered":
        return datasets.gaussian_mixture3d(n, components=6, sigma=0.05, seed=6)
    if name == "planar":
        return datasets.pad_to_3d(datasets.taxi_like2d(n, components=8, seed=7))
    raise ValueError(name)


def main():
    engine, n, k, name, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    use_gpu = engine == "hip"
    # DIST_BACKEND=gloo (default): host-staged messages -- CPU ranks, or ranks sharing one GPU.  DIST_BACKEND=nccl: RCCL, device
    # tensors in every collective and point-to-point message, one rank per GPU (bench.py's way; on a one-GPU box: one rank)
    backend = os.environ.get("DIST_BACKEND", "gloo")
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
        dev = torch.device("cuda", 0) if use_gpu else torch.device("cpu")
        if use_gpu:
            torch.cuda.set_device(0)
    rank, world = dist.get_rank(), dist.get_world_size()
    pts = make_points(name, n)
    lo, hi = n * rank // world, n * (rank + 1) // world  # arbitrary initial ownership: contiguous slices
    solver = tkd.ShardedTrueKNN(dev, engine_factory=None if use_gpu else CheckerEngine, halo_levels=int(os.environ.get("HALO_LEVELS", "1")))
    if name == "counter":
        solver.load_counter_based(n, seed=0)  # what `bench.py --gpus N --scaling strong` does: every rank draws its slice
    else:
        solver.load_points(torch.from_numpy(pts[lo:hi]), torch.arange(lo, hi, dtype=torch.int32))
    if os.environ.get("DBSCAN_EPS"):
        if os.environ.get("DBSCAN_MAX_NOISE"):  # the auto-grown eps: DBSCAN_EPS is eps0
            r = solver.dbscan_auto(float(os.environ["DBSCAN_EPS"]), int(os.environ.get("DBSCAN_MINPTS", "4")),
                                   max_noise=float(os.environ["DBSCAN_MAX_NOISE"]))
        else:
            r = solver.dbscan(float(os.environ["DBSCAN_EPS"]), int(os.environ.get("DBSCAN_MINPTS", "4")))
        rows = torch.cat([solver.ids.view(-1, 1).double().cpu(), r["labels"].view(-1, 1).double().cpu(),
                          r["core"].view(-1, 1).double().cpu()], dim=1).to(dev)
        every = int(os.environ.get("GATHER_EVERY", "1"))
        if every > 1:  # a run at size: the rows of the ids divisible by GATHER_EVERY, to rank 0 only
            rows = rows[(solver.ids % every == 0).to(rows.device)]
            got = torch.cat(solver.comm.exchange_rows([rows if p == 0 else rows[:0] for p in range(world)], 3, torch.float64, dev), dim=0).cpu().numpy()
        else:
            got = torch.cat(solver.comm.exchange_rows([rows for _ in range(world)], 3, torch.float64, dev), dim=0).cpu().numpy()
        got = got[np.argsort(got[:, 0])]
        print("rank %d tile=%d halo=%d clusters=%d label rounds=%d" % (rank, len(solver.points), r["info"]["halo_points"],
                                                                    r["info"]["clusters"], r["info"]["rounds"]), flush=True)
        if rank == 0:
            np.savez(out, gids=got[:, 0].astype(np.int64), labels=got[:, 1].astype(np.int32), core=got[:, 2].astype(bool),
                     clusters=r["info"]["clusters"], eps=r["info"].get("eps", 0.0), eps_rounds=r["info"].get("rounds", 0),
                     noise=r["info"].get("noise", -1))
        dist.barrier()
        dist.destroy_process_group()
        return
    r0 = float(os.environ.get("START_RADIUS", datasets.start_radius(n, k)))
    import time
    t0 = time.perf_counter()
    info = solver.solve(k, r0)
    if use_gpu:
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    info = solver.solve(k, r0)
    if use_gpu:
        torch.cuda.synchronize()
    print('rank %d tile=%d halo=%d exchanges=%d solve_s first=%.4f second=%.4f kernel_ms=%.2f' % (rank, len(solver.points), info['halo_points'], info['halo_exchanges'], t1 - t0, time.perf_counter() - t1, info['dominant_kernel_ms']), flush=True)
    every = int(os.environ.get("GATHER_EVERY", "1"))
    if every > 1:
        # a run at size: only the rows of the ids divisible by GATHER_EVERY travel to rank 0 (all of them would be gigabytes)
        keep = torch.nonzero(solver.ids % every == 0).flatten()
        rows = torch.cat([solver.ids[keep].view(-1, 1).double(), solver.last["idx"][keep].double(), solver.last["dist"][keep].double(),
                          solver.last["intersections"][keep].view(-1, 1).double()], dim=1)
        got = torch.cat(solver.comm.exchange_rows([rows if p == 0 else rows[:0] for p in range(world)], rows.shape[1], torch.float64, dev), dim=0).cpu().numpy()
        got = got[np.argsort(got[:, 0])]
        gids, idx, dst, isect = (got[:, 0].astype(np.int64), got[:, 1:1 + k].astype(np.int32), got[:, 1 + k:1 + 2 * k].astype(np.float32),
                                 got[:, 1 + 2 * k].astype(np.int64))
    else:
        gids, idx, dst, isect = solver.gather_rows()
    if rank == 0:
        np.savez(out, gids=gids, idx=idx, dist=dst, isect=isect, rounds=info["rounds"],
                 exchanges=info["halo_exchanges"], halo_points=info["halo_points"], tile=len(solver.points),
                 halo_by_exchange=np.asarray(info["halo_points_by_exchange"], np.int64), total_isect=info["total_intersections"],
                 one_pass=bool(info.get("halo_select_one_pass", False)))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
