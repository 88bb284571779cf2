#!/usr/bin/env python3
"""Developer probe: phase times of the sharded solve with several ranks SHARING one GPU (gloo,
host-staged messages -- the exchange time is not RCCL's; the selection / halo build / solve are).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 \
        --master-port 29611 scripts/sharded_probe.py 8000000 10
"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from owlraytracing_amd import datasets, distributed as tkd  # noqa: E402


def main():
    n_total, k = int(sys.argv[1]), int(sys.argv[2])
    os.environ["TKNN_SHARD_PROFILE"] = "1"
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    solver = tkd.ShardedTrueKNN(dev, halo_levels=(int(os.environ["HALO_LEVELS"]) if "HALO_LEVELS" in os.environ else None))
    solver.load_counter_based(n_total, seed=0)
    r0 = datasets.start_radius(n_total, k)
    for it in range(3):
        dist.barrier()
        info = solver.solve(k, r0)
        if it == 2:
            ph = {kk: round(v, 2) for kk, v in info["phase_ms"].items()}
            print("rank %d: n_local=%d halo=%d exchanges=%d rounds=%d phases_ms=%s kernel_ms=%.2f" % (
                dist.get_rank(), len(solver.points), info["halo_points"], info["halo_exchanges"], info["rounds"], ph,
                info["dominant_kernel_ms"]), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
