"""Multi-GPU TrueKNN: Morton tiles + halo exchange.

The reference has no multi-GPU path for this workload (it replicates every buffer and the BVH on
every device and pins the sample to one GPU, owl/RayGen.cpp:150-200, hostCode.cpp:141).  Here the
path shards naturally -- queries are independent -- with ONE real exchange step:

  * the global point set is cut into W contiguous ranges of the 63-bit Morton order with equal
    counts (splitters from a sorted sample); rank g owns the queries and result rows of tile g and
    keeps an LBVH over its own points (built once, like the single-GPU build); because a Morton
    range is not spatially compact (Z-curve jumps), a tile is described by the bounding boxes of its
    level-2 Morton cells (<= 64 compact boxes), not by one box;
  * per solve, every rank sends each peer the points of its tile that lie within the halo radius of
    one of the peer's cell boxes -- exactly the foreign points that can be box candidates of the
    peer's queries (the reference's candidate test is an L-inf box) -- as one
    point-to-point exchange (RCCL send/recv over xGMI; a few MB per pair, latency-bound);
  * the engine searches own tree + halo tree; a query's row is exact as soon as its final radius
    does not exceed the halo radius, which one 8-byte all-reduce checks; stragglers double the halo
    radius and go again.  Neighbour indices are global, so rows equal the single-GPU result.

One process per GPU (torch.distributed, backend "nccl" = RCCL on ROCm); with backend "gloo" the
same code runs with host-staged messages (CPU tests, or several ranks sharing one GPU).
"""
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from . import _lib, datasets


def morton63(points, lo, extent):
    """63-bit Morton codes (21 bits per axis, x most significant) of float32 points, as int64."""
    scale = 2097151.0 / extent if extent > 0 else 0.0
    q = ((points.double() - lo.double()) * scale).clamp_(0, 2097151).long()

    def spread(v):
        v = v & 0x1FFFFF
        v = (v | (v << 32)) & 0x1F00000000FFFF
        v = (v | (v << 16)) & 0x1F0000FF0000FF
        v = (v | (v << 8)) & 0x100F00F00F00F00F
        v = (v | (v << 4)) & 0x10C30C30C30C30C3
        v = (v | (v << 2)) & 0x1249249249249249
        return v

    return (spread(q[:, 0]) << 2) | (spread(q[:, 1]) << 1) | spread(q[:, 2])


def sample_positions(count, samples, device):
    """`samples` evenly spaced positions in [0, count-1], first and last included, in exact integer
    arithmetic (a float32 linspace rounds its end point up to `count` once count > 2**24: an
    out-of-bounds gather on tiles of 50 M points)."""
    last = max(int(count) - 1, 0)
    steps = torch.arange(samples, dtype=torch.int64, device=device)
    return (steps * last) // max(samples - 1, 1)


class _Comm:
    """The three collectives the path needs, on device tensors (nccl) or host-staged (gloo)."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.staged = dist.get_backend(group) != "nccl"
        self._caps = {}  # (tag, "in" | "out") -> rows per peer the next fixed-capacity exchange under that tag holds

    def _wire(self, t):
        return t.cpu() if self.staged and t.is_cuda else t

    def all_reduce(self, t, op):
        w = self._wire(t)
        dist.all_reduce(w, op=op, group=self.group)
        if w is not t:
            t.copy_(w)
        return t

    def all_gather(self, t):
        w = self._wire(t.contiguous())
        out = [torch.empty_like(w) for _ in range(self.world)]
        dist.all_gather(out, w, group=self.group)
        return torch.stack(out).to(t.device)

    def exchange_rows(self, rows_per_peer, width, dtype, device, tag=None, counts_in=None):
        """all-to-all-v of 2-D row blocks: rows_per_peer[p] goes to rank p; returns the list of
        blocks received (index = source rank).

        Without ``tag``: counts first (an all-gather and a host round trip), then one batched
        send/recv of exactly those rows.  With ``counts_in`` (rows per source, known to the caller):
        just that send/recv.  With ``tag`` (the per-step halo exchange): ONE batched
        send/recv of fixed-capacity messages whose first row carries the count -- the capacity of
        every ordered pair is what both ends remember of the pair's last exchange under that tag
        (x 1.25 + 64 rows); the first exchange under a tag, and any pair whose rows outgrow the
        capacity (both ends see it: the sender by its rows, the receiver by the header), go
        through exact-size messages once more.  Float32 headers hold counts < 2**24 exactly."""
        world, me = self.world, self.rank
        if world == 1:  # nobody to exchange with (and no count to agree on)
            return [rows_per_peer[0]]
        if tag is not None and dtype == torch.float32 and (tag, "in") in self._caps:
            cap_in, cap_out = self._caps[(tag, "in")], self._caps[(tag, "out")]
            recv = [None] * world
            ops = []
            msgs = []
            for p in range(world):
                if p == me:
                    continue
                rows = rows_per_peer[p]
                m = torch.zeros((cap_out[p] + 1, width), dtype=dtype, device=rows.device)
                m[0, 0] = float(len(rows) % (1 << 24))  # (two cells: float32 holds integers below 2**24 exactly)
                m[0, 1] = float(len(rows) >> 24)
                take = min(len(rows), cap_out[p])
                if take:
                    m[1:1 + take] = rows[:take]
                msgs.append(m)
                ops.append(dist.P2POp(dist.isend, self._wire(m), p, group=self.group))
                recv[p] = torch.empty((cap_in[p] + 1, width), dtype=dtype, device="cpu" if self.staged else device)
                ops.append(dist.P2POp(dist.irecv, recv[p], p, group=self.group))
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            peers = [p for p in range(world) if p != me]
            heads = torch.stack([recv[p][0, :2] for p in peers]).tolist() if peers else []  # the step's one host round trip
            counts_in = [0] * world
            for p, h in zip(peers, heads):
                counts_in[p] = int(h[0]) + (int(h[1]) << 24)
            out = [None] * world
            again = []  # pairs whose rows did not fit: exact-size messages, both ends know
            for p in range(world):
                if p == me:
                    out[p] = rows_per_peer[p]
                    continue
                if counts_in[p] > cap_in[p]:
                    again.append(("in", p))
                else:
                    out[p] = recv[p][1:1 + counts_in[p]].to(device)
                if len(rows_per_peer[p]) > cap_out[p]:
                    again.append(("out", p))
            if again:
                ops, late = [], {}
                for kind, p in again:
                    if kind == "out":
                        ops.append(dist.P2POp(dist.isend, self._wire(rows_per_peer[p].contiguous()), p, group=self.group))
                    else:
                        late[p] = torch.empty((counts_in[p], width), dtype=dtype, device="cpu" if self.staged else device)
                        ops.append(dist.P2POp(dist.irecv, late[p], p, group=self.group))
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
                for p, t in late.items():
                    out[p] = t.to(device)
            self._remember(tag, counts_in, [len(r) for r in rows_per_peer])
            return out
        if counts_in is None:
            counts = torch.tensor([len(r) for r in rows_per_peer], dtype=torch.int64, device=device)
            counts_in = self.all_gather(counts)[:, self.rank].tolist()  # [src] rows coming from src
        # (else: the caller knows what arrives -- the answers to rows it has sent, say -- and no count travels)
        recv = [torch.empty((int(c), width), dtype=dtype, device="cpu" if self.staged else device) for c in counts_in]
        ops = []
        for p in range(self.world):
            if p == self.rank:
                continue
            if len(rows_per_peer[p]):
                ops.append(dist.P2POp(dist.isend, self._wire(rows_per_peer[p].contiguous()), p, group=self.group))
            if counts_in[p]:
                ops.append(dist.P2POp(dist.irecv, recv[p], p, group=self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        recv[self.rank] = rows_per_peer[self.rank]
        if tag is not None and dtype == torch.float32:
            self._remember(tag, [int(c) for c in counts_in], [len(r) for r in rows_per_peer])
        return [r.to(device) for r in recv]

    def has_caps(self, tag):
        return (tag, "in") in self._caps

    def caps_out(self, tag):
        return list(self._caps[(tag, "out")])

    def exchange_fixed(self, messages, starts, counts_dev, tag, device, exact_blocks):
        """The tagged exchange for rows that were SELECTED straight into fixed-capacity messages (TrueKNN.halo_select_fixed): the
        messages leave as they are, the headers of what arrives and my own counts come to the host in ONE read -- the step's only
        round trip before the solve (the two-pass selection had one of its own).  A pair whose rows outgrew its capacity goes
        through exact-size messages once more; ``exact_blocks()`` then returns the per-peer rows the two-pass selection gives.
        Returns (rows received per source, rows sent per destination)."""
        world, me = self.world, self.rank
        cap_in, cap_out = self._caps[(tag, "in")], self._caps[(tag, "out")]
        recv, ops = [None] * world, []
        for p in range(world):
            if p == me:
                continue
            m = messages[starts[p]: starts[p] + cap_out[p] + 1]
            ops.append(dist.P2POp(dist.isend, self._wire(m), p, group=self.group))
            recv[p] = torch.empty((cap_in[p] + 1, 4), dtype=torch.float32, device="cpu" if self.staged else device)
            ops.append(dist.P2POp(dist.irecv, recv[p], p, group=self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        peers = [p for p in range(world) if p != me]
        mine = counts_dev.to(recv[peers[0]].device if peers else counts_dev.device).double()
        heads = [recv[p][0, :2].double() for p in peers]
        flat = torch.cat([mine] + heads).tolist()  # the one host round trip: my counts, then (low, high) per source
        counts_out = [int(c) for c in flat[:world]]
        counts_in = [0] * world
        for j, p in enumerate(peers):
            counts_in[p] = int(flat[world + 2 * j]) + (int(flat[world + 2 * j + 1]) << 24)
        sent = [messages[starts[p] + 1: starts[p] + 1 + min(counts_out[p], cap_out[p])] for p in range(world)]
        out = [None] * world
        again_in = [p for p in peers if counts_in[p] > cap_in[p]]
        again_out = [p for p in peers if counts_out[p] > cap_out[p]]
        for p in range(world):
            if p == me:
                out[p] = messages[:0]
            elif p not in again_in:
                out[p] = recv[p][1:1 + counts_in[p]].to(device)
        if again_in or again_out:
            exact = exact_blocks() if again_out else None
            ops, late = [], {}
            for p in again_out:
                sent[p] = exact[p]
                ops.append(dist.P2POp(dist.isend, self._wire(exact[p].contiguous()), p, group=self.group))
            for p in again_in:
                late[p] = torch.empty((counts_in[p], 4), dtype=torch.float32, device="cpu" if self.staged else device)
                ops.append(dist.P2POp(dist.irecv, late[p], p, group=self.group))
            for req in dist.batch_isend_irecv(ops):
                req.wait()
            for p, t in late.items():
                out[p] = t.to(device)
        self._remember(tag, counts_in, counts_out)
        return out, sent

    def _remember(self, tag, counts_in, counts_out):
        """capacities of the next exchange under ``tag``: both ends of a pair derive the same number from the same count"""
        grow = lambda c: int(c) + int(c) // 4 + 64  # noqa: E731
        old_in, old_out = self._caps.get((tag, "in")), self._caps.get((tag, "out"))
        self._caps[(tag, "in")] = [max(grow(c), old_in[p] if old_in else 0) for p, c in enumerate(counts_in)]
        self._caps[(tag, "out")] = [max(grow(c), old_out[p] if old_out else 0) for p, c in enumerate(counts_out)]


class ShardedTrueKNN:
    """TrueKNN over a point set sharded across the ranks of a process group.

    ``engine_factory(device)`` must return an object with the ``owlraytracing_amd.trueknn.TrueKNN``
    interface (build / set_halo / solve); the default is that class, i.e. the HIP engine.  Tests
    inject a checker-backed stand-in to run the partition / halo / validity logic on CPU ranks.
    """

    def __init__(self, device, kernel=_lib.KERNEL_AUTO, engine_factory=None, group=None, halo_levels=None):
        self.device = torch.device(device)
        self.kernel = kernel
        self.comm = _Comm(group)
        if engine_factory is None:
            from .trueknn import TrueKNN
            engine_factory = lambda dev: TrueKNN(device=dev.index)  # noqa: E731
        self.engine = engine_factory(self.device)
        self._engine_factory = engine_factory
        self.db_engine = None   # second engine over own + halo points (dbscan)
        self._db_cache = None   # (eps, what _db_setup returned for it)
        # first halo radius = start_radius * 2**halo_levels; None: from the density of the set, the level
        # at which a box is expected to hold 32 k points (nearly every query has finished by then, so
        # one exchange and one solve do; stragglers still widen the halo and go again)
        self.halo_levels = halo_levels
        self.extent = None      # (3,) float64 extents of the whole set
        self.points = None      # (m,3) float32 owned points, on self.device
        self.ids = None         # (m,) int32 global ids of the owned points
        self.tile_boxes = None  # (W, MAX_CELLS, 6) float64 on the host: boxes of every rank's Morton cells
        self.cell_boxes = None  # (C,6) my own cells
        self.cell_slices = None  # [(start, end)] of my cells in self.points
        self.n_total = 0
        self.last = None
        self.profile = bool(os.environ.get("TKNN_SHARD_PROFILE"))  # per-phase wall times in info (adds syncs)
        # Opt-in (TKNN_SHARD_OVERLAP=1): solve the queries no foreign point can reach ("interior": not inside any peer's
        # widened cell box) in the own tree while the halo travels and its tree is built; the boundary queries follow
        # with own + halo tree (tknnSolveOptions.phase).  Rows are identical either way (tests).  Off by default: the
        # packet kernel is persistent and fills every CU, so the exchange's own kernels (RCCL send/recv, the halo
        # tree build) queue behind it instead of running beside it, and two launches + two host round trips replace
        # one -- measured with two ranks sharing one MI355X (gloo-staged messages): 12.2 ms per step against 6.5 ms
        # in the serial order.
        self.overlap = os.environ.get("TKNN_SHARD_OVERLAP", "0") == "1"
        self._boundary_marked = False
        self._out = {}

    # ---- one-time distribution -------------------------------------------------------------
    def load_counter_based(self, n_total, seed=0):
        """Every rank generates its slice of the C4-style counter-based uniform set, then tiles."""
        w, r = self.comm.world, self.comm.rank
        lo, hi = n_total * r // w, n_total * (r + 1) // w
        pts = torch.from_numpy(datasets.uniform3d_counter(lo, hi, seed=seed)).to(self.device)
        ids = torch.arange(lo, hi, dtype=torch.int32, device=self.device)
        self.load_points(pts, ids)

    def load_points(self, points, ids):
        """Redistribute (points, global ids) so that rank g holds Morton tile g; build own tree."""
        comm, dev = self.comm, self.device
        if isinstance(points, np.ndarray):
            points = torch.from_numpy(datasets.pad_to_3d(points))
        if isinstance(ids, np.ndarray):
            ids = torch.from_numpy(np.ascontiguousarray(ids, dtype=np.int32))
        points, ids = points.to(dev).float().contiguous(), ids.to(dev).int().contiguous()
        big = torch.finfo(torch.float32).max
        lo = points.min(0).values.double() if len(points) else torch.full((3,), big, dtype=torch.float64, device=dev)
        hi = points.max(0).values.double() if len(points) else torch.full((3,), -big, dtype=torch.float64, device=dev)
        comm.all_reduce(lo, dist.ReduceOp.MIN)
        comm.all_reduce(hi, dist.ReduceOp.MAX)
        extent = float((hi - lo).max())
        self.extent = (hi - lo).cpu()
        codes = morton63(points, lo, extent)
        order = torch.argsort(codes)
        codes, points, ids = codes[order], points[order], ids[order]
        # splitters: W-quantiles of the union of per-rank sorted samples
        s = 1024
        pick = sample_positions(len(codes), s, dev)
        sample = codes[pick] if len(codes) else torch.full((s,), 2 ** 62, dtype=torch.int64, device=dev)
        allsamp = comm.all_gather(sample).flatten().sort().values
        cut = torch.tensor([len(allsamp) * (g + 1) // comm.world for g in range(comm.world - 1)], device=dev).long()
        splitters = allsamp[cut] if comm.world > 1 else allsamp[:0]
        dest = torch.searchsorted(splitters, codes, right=True)
        rows = torch.cat([points, ids.view(torch.float32).unsqueeze(1)], dim=1)  # 16-byte rows: x y z id-bits
        blocks = [rows[dest == p] for p in range(comm.world)]
        got = torch.cat(comm.exchange_rows(blocks, 4, torch.float32, dev), dim=0)
        if len(got) == 0:
            raise RuntimeError("rank %d received an empty tile; use fewer ranks for this point set" % comm.rank)
        # keep the tile in Morton order: its cells are then contiguous slices
        mycodes = morton63(got[:, :3], lo, extent)
        order = torch.argsort(mycodes)
        got, mycodes = got[order].contiguous(), mycodes[order]
        self.points = got[:, :3].contiguous()
        self._db_cache = None
        self.ids = got[:, 3].contiguous().view(torch.int32)
        self._rows = got  # 16-byte wire rows: x y z id-bits
        n_local = torch.tensor([len(self.points)], dtype=torch.int64, device=dev)
        self.n_total = int(comm.all_reduce(n_local.clone(), dist.ReduceOp.SUM).item())
        # A Morton range is NOT spatially compact: a tile that ends just past a jump of the Z curve
        # has a bounding box spanning far-apart regions, and "points near the tile's box" would be
        # most of the data set.  So a tile is described by the boxes of its level-2 Morton cells
        # (<= 64 per tile, each inside one cell of the 4x4x4 grid, hence compact).
        cells = mycodes >> (63 - 3 * self.CELL_LEVEL)
        uniq, counts = torch.unique_consecutive(cells, return_counts=True)
        ends = torch.cumsum(counts, 0).tolist()
        starts = [0] + ends[:-1]
        self.cell_slices = list(zip(starts, ends))
        boxes = torch.full((self.MAX_CELLS, 6), float("inf"), dtype=torch.float64, device=dev)
        boxes[:, 3:] = -float("inf")
        for j, (s0, s1) in enumerate(self.cell_slices):
            seg = self.points[s0:s1]
            boxes[j, :3] = seg.min(0).values.double()
            boxes[j, 3:] = seg.max(0).values.double()
        self.cell_boxes = boxes[: len(self.cell_slices)].cpu()
        self.tile_boxes = comm.all_gather(boxes).cpu()  # (W, MAX_CELLS, 6); empty slots are inverted boxes
        self.engine.build(self.points, self.ids)

    CELL_LEVEL = 2
    MAX_CELLS = 64

    # ---- per-solve exchange ---------------------------------------------------------------------
    def _halo_blocks(self, radius):
        """For each peer, the rows of my tile within `radius` (L-inf) of any of the peer's cell boxes:
        exactly the foreign points that can be box candidates of the peer's queries.

        Per peer: my cells whose box meets one of the peer's widened boxes are found on the host
        (<= 64 x 64 tiny tests); only those slices are tested point by point, in float32 against
        boxes widened in float64 (radius, a relative 1e-5, an ulp-of-coordinate margin) and rounded
        OUTWARD, so the selection can only err on the side of sending a point too many."""
        comm, dev = self.comm, self.device
        reach = float(radius) * (1.0 + 1e-5) + 1e-30
        if hasattr(self.engine, "halo_select"):
            # the engine's kernels do the selection on its own Morton-sorted points: widened peer boxes
            # in, per-peer contiguous wire rows out (tknnHaloSelect)
            boxes, owner = self._peer_boxes(reach)
            if boxes is None:
                return [self._rows[:0] for _ in range(comm.world)]
            rows, counts = self.engine.halo_select(boxes, owner, comm.world)
            self._boundary_marked = True  # the count pass has marked my points inside a peer's box: my boundary queries
            return list(torch.split(rows, counts))
        rows = self._rows
        blocks = []
        mine = self.cell_boxes  # (C,6) float64, host
        for peer in range(comm.world):
            if peer == comm.rank:
                blocks.append(rows[:0])
                continue
            pb = self.tile_boxes[peer]
            pb = pb[pb[:, 0] <= pb[:, 3]]  # drop empty slots
            mag = pb.abs().max(dim=1, keepdim=True).values
            lo64, hi64 = pb[:, :3] - reach - 1e-6 * mag, pb[:, 3:] + reach + 1e-6 * mag
            # which of my cells meet any widened peer box (host)
            meet = ((mine[:, None, :3] <= hi64[None]) & (mine[:, None, 3:] >= lo64[None])).all(dim=2)  # (C,P)
            lo, hi = self._widen(pb, reach)
            lo, hi = lo.to(dev), hi.to(dev)
            picked = []
            for c, (s0, s1) in enumerate(self.cell_slices):
                which = torch.nonzero(meet[c]).flatten()
                if len(which) == 0:
                    continue
                seg = self.points[s0:s1]
                blo, bhi = lo[which.to(dev)], hi[which.to(dev)]
                inside = ((seg[:, None, :] >= blo[None]) & (seg[:, None, :] <= bhi[None])).all(dim=2).any(dim=1)
                picked.append(rows[s0:s1][inside])
            blocks.append(torch.cat(picked, dim=0) if picked else rows[:0])
        return blocks

    def _peer_boxes(self, reach):
        """(boxes (m,6) float32, owner (m,) int32) of every peer's Morton cells widened by ``reach``, or (None, None)"""
        comm = self.comm
        boxes, owner = [], []
        for peer in range(comm.world):
            if peer == comm.rank:
                continue
            pb = self.tile_boxes[peer]
            pb = pb[pb[:, 0] <= pb[:, 3]]  # drop empty slots
            if len(pb) == 0:
                continue
            lo, hi = self._widen(pb, reach)
            boxes.append(torch.cat([lo, hi], dim=1))
            owner.append(torch.full((len(pb),), peer, dtype=torch.int32))
        if not boxes:
            return None, None
        return torch.cat(boxes), torch.cat(owner)

    @staticmethod
    def _widen(pb, reach):
        """float32 boxes containing pb (float64 cell boxes) widened by `reach`, a relative 1e-6 and
        rounded OUTWARD: the selection can only err on the side of sending a point too many."""
        mag = pb.abs().max(dim=1, keepdim=True).values
        lo64, hi64 = pb[:, :3] - reach - 1e-6 * mag, pb[:, 3:] + reach + 1e-6 * mag
        lo, hi = lo64.float(), hi64.float()
        lo = torch.where(lo.double() > lo64, torch.nextafter(lo, torch.full_like(lo, -float("inf"))), lo)
        hi = torch.where(hi.double() < hi64, torch.nextafter(hi, torch.full_like(hi, float("inf"))), hi)
        return lo, hi

    @staticmethod
    def _merge_infos(a, b, levels_counted_twice):
        """info of two engine calls that together are one solve (interior + boundary queries; a solve and the re-solve of
        the queries it left unfinished, whose ``levels_counted_twice`` active levels the first call has counted already)"""
        merged = dict(b)
        for key in ("total_intersections", "total_active_rounds", "node_tests", "point_tests", "tie_rows", "tie_rows_left", "solve_ms", "tie_ms"):
            merged[key] = a[key] + b[key]
        merged["total_active_rounds"] -= levels_counted_twice
        merged["unfinished"] = b["unfinished"] if levels_counted_twice else a["unfinished"] + b["unfinished"]
        merged["rounds"] = max(a["rounds"], b["rounds"])
        merged["final_radius"] = max(a["final_radius"], b["final_radius"])
        merged["dominant_kernel_launches"] = a["dominant_kernel_launches"] + b["dominant_kernel_launches"]
        merged["dominant_kernel_ms"] = ((a["dominant_kernel_ms"] * a["dominant_kernel_launches"] + b["dominant_kernel_ms"] * b["dominant_kernel_launches"])
                                        / max(merged["dominant_kernel_launches"], 1))
        return merged

    def solve(self, k, start_radius, max_rounds=64, want_fb=False):
        comm, dev = self.comm, self.device
        r0 = np.float32(start_radius)
        level_cap = self.halo_levels
        if level_cap is None:
            ext = [float(e) for e in self.extent if float(e) > 0]
            measure = float(np.prod(ext)) if ext else 0.0
            level_cap = 0
            if measure > 0:
                density = self.n_total / measure
                while level_cap < 6 and density * (2.0 * float(r0) * 2 ** level_cap) ** len(ext) < 32.0 * k:
                    level_cap += 1
        exchanges, halo_points = 0, 0
        phase = {"select": 0.0, "exchange": 0.0, "halo_build": 0.0, "solve": 0.0, "reduce": 0.0}
        profile = self.profile

        def lap(name, t0):
            if profile:
                if dev.type == "cuda":
                    torch.cuda.synchronize(dev)
                phase[name] += (time.perf_counter() - t0) * 1e3
            return time.perf_counter()

        first = True
        one_pass = False  # the first exchange of this solve selected its rows straight into fixed-capacity messages
        overlapped = False
        halo = None                         # every foreign row received so far (the halo tree's points)
        sent = [None] * comm.world          # ids of my rows each peer holds already
        round_halo = []                     # rows received per exchange
        res = None
        while True:
            halo_radius = np.float32(r0)
            for _ in range(level_cap):
                halo_radius = np.float32(halo_radius * np.float32(2))
            t = time.perf_counter()
            self._boundary_marked = False
            tag = ("halo", level_cap, first)
            # The step's usual form (round 4): both ends of every pair remember how much travelled under this tag, so the rows are
            # selected straight into messages of that capacity (one pass, no count pass, no host round trip of its own) and the
            # exchange's one read of the headers also brings my own counts.  The first exchange under a tag, the straggler rounds
            # (shells) and engines without the one-pass selection go the two-pass way below.
            fixed = (first and comm.world > 1 and comm.has_caps(tag) and hasattr(self.engine, "halo_select_fixed")
                     and not (self.overlap and getattr(self.engine, "supports_phases", False)))
            fixed_sel = None
            if fixed:
                reach = float(halo_radius) * (1.0 + 1e-5) + 1e-30
                boxes, owner = self._peer_boxes(reach)
                if boxes is None:
                    fixed = False
                else:
                    fixed_sel = self.engine.halo_select_fixed(boxes, owner, comm.world, [0 if p == comm.rank else c for p, c in enumerate(comm.caps_out(tag))])
                    self._boundary_marked = True
                    one_pass = True
            blocks = self._halo_blocks(halo_radius) if not fixed else None
            if not first:
                # a straggler round: the peers hold what the smaller radius selected -- only the SHELL between the two
                # radii travels ("only the incremental shell (r_{t-1}, r_t] need be sent after round 1")
                for p in range(comm.world):
                    if p != comm.rank and len(blocks[p]) and sent[p] is not None and len(sent[p]):
                        ids_p = blocks[p][:, 3].contiguous().view(torch.int32)
                        blocks[p] = blocks[p][~torch.isin(ids_p, sent[p])]
            if not fixed:
                for p in range(comm.world):
                    if p != comm.rank and len(blocks[p]):
                        ids_p = blocks[p][:, 3].contiguous().view(torch.int32)
                        sent[p] = ids_p if sent[p] is None else torch.cat([sent[p], ids_p])
            t = lap("select", t)
            two_phases = (first and self.overlap and self._boundary_marked and getattr(self.engine, "supports_phases", False)
                          and dev.type == "cuda" and k <= 64 and self.kernel in (_lib.KERNEL_AUTO, _lib.KERNEL_TEAM))
            solve_kw = dict(kernel=self.kernel, max_rounds=level_cap + 1, want_fb=want_fb, want_levels=True, allow_unfinished=True)
            interior = {}
            worker = None
            overlapped = overlapped or two_phases
            if two_phases:
                # interior queries: own tree only, on a stream and a host thread of their own (the call returns when
                # its kernels are done, so it cannot be issued from the thread that drives the exchange)
                import threading

                n_own = len(self.points)
                if self._out.get("idx") is None or tuple(self._out["idx"].shape) != (n_own, k):
                    self._out = {"idx": torch.empty((n_own, k), dtype=torch.int32, device=dev),
                                 "dist": torch.empty((n_own, k), dtype=torch.float32, device=dev),
                                 "intersections": torch.empty((n_own,), dtype=torch.int64, device=dev),
                                 "levels": torch.empty((n_own,), dtype=torch.int32, device=dev)}
                if want_fb and self._out.get("fb") is None:
                    self._out["fb"] = torch.empty((n_own * k * 24,), dtype=torch.uint8, device=dev)
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                self.engine.set_halo(None, None)

                def run_interior():
                    try:
                        interior["res"] = self.engine.solve(k, float(r0), out=self._out, phase=1, stream=side, **solve_kw)
                    except BaseException as e:  # re-raised on the driving thread
                        interior["error"] = e

                worker = threading.Thread(target=run_interior, name="tknn-interior")
                t_int = time.perf_counter()
                worker.start()
            try:
                # (tagged: fixed-capacity messages with the count in their first row -- one exchange, one host round
                # trip -- once a first exchange at this radius level has told both ends of every pair how much travels)
                if fixed:
                    got, went = comm.exchange_fixed(fixed_sel[0], fixed_sel[1], fixed_sel[2], tag, dev, lambda: self._halo_blocks(halo_radius))
                    for p in range(comm.world):  # what each peer holds of mine now (the straggler rounds send shells)
                        if p != comm.rank and len(went[p]):
                            sent[p] = went[p][:, 3].contiguous().view(torch.int32)
                else:
                    got = comm.exchange_rows(blocks, 4, torch.float32, dev, tag=tag)
                got[comm.rank] = got[comm.rank][:0]
                new_rows = torch.cat(got, dim=0)
                t = lap("exchange", t)
                exchanges += 1
                round_halo.append(int(len(new_rows)))
                halo = new_rows if halo is None else torch.cat([halo, new_rows], dim=0)
                halo_points = len(halo)
                self.engine.set_halo(halo[:, :3].contiguous(), halo[:, 3].contiguous().view(torch.int32))
                t = lap("halo_build", t)
            finally:
                # the interior solve uses self.engine on its own stream: never unwind past it
                if worker is not None:
                    worker.join()
            if worker is not None:
                if "error" in interior:
                    raise interior["error"]
                phase["interior_solve_overlapped"] = phase.get("interior_solve_overlapped", 0.0) + (time.perf_counter() - t_int) * 1e3
                t = time.perf_counter()
                # boundary queries: own + halo tree, rows and levels of the interior queries stay
                res = self.engine.solve(k, float(r0), out={kk: v for kk, v in interior["res"].items() if kk != "info"}, phase=2, **solve_kw)
                res["info"] = self._merge_infos(interior["res"]["info"], res["info"], 0)
            elif first:
                # levels 0..level_cap are exact with this halo; stop there and see who is left
                res = self.engine.solve(k, float(r0), **solve_kw)
            else:
                # only the queries the rounds so far left unfinished, over own + widened halo tree; every other row stays
                before = res["info"]
                res = self.engine.solve(k, float(r0), out={kk: v for kk, v in res.items() if kk != "info"}, phase=3, **solve_kw)
                res["info"] = self._merge_infos(before, res["info"], int(before["unfinished"]) * level_cap)
            t = lap("solve", t)
            first = False
            # ONE all-reduce per solve step: is anybody left (a maximum says so as well as a sum), and the rounds the ranks took
            left = torch.tensor([int(res["info"]["unfinished"]), int(res["info"]["rounds"])], dtype=torch.int64, device=dev)
            comm.all_reduce(left, dist.ReduceOp.MAX)
            left_host = left.tolist()
            done = int(left_host[0]) == 0
            rounds_all = int(left_host[1])
            t = lap("reduce", t)
            if done:
                break
            if level_cap + 1 >= max_rounds:
                raise _lib.TknnError(-4, "max_rounds reached with unfinished queries")
            level_cap += 1  # stragglers need the next radius level: widen the halo by its shell and solve them again
        info = dict(res["info"])
        info["rounds"] = rounds_all
        info["halo_exchanges"] = exchanges
        info["halo_points"] = halo_points
        info["halo_points_by_exchange"] = round_halo  # the first exchange's halo, then the shells of the straggler rounds
        info["halo_levels"] = level_cap
        info["overlapped"] = bool(overlapped)
        info["halo_select_one_pass"] = bool(one_pass)
        if profile:
            info["phase_ms"] = phase
        self.last = res
        self.last["info"] = info
        return info

    # ---- RT-DBSCAN over the tiles (last row) -----------------------------------
    def dbscan(self, eps, min_pts, _reuse_setup=False):
        """DBSCAN of the whole set with the single-GPU spec (oracle/dbscan_oracle.c): returns
        dict(labels (m,) int32, core (m,) bool) for the points this rank owns (``self.ids`` order) and
        info(clusters, rounds, halo_points).  Clusters are numbered by ascending smallest core id, so
        the labels equal a single-process run over the union.

        Each rank clusters its tile plus a halo of radius 2 eps: core flags are then exact for its own
        points and for every foreign point that can be their neighbour, and every edge of the
        neighbour graph is seen by a rank owning one of its ends.  Components are merged by label
        propagation: label = smallest core id known for the component; a round takes the minimum
        over each local component, sends the labels of halo copies back to their owners (minimum) and
        the owners' labels out to the copies again, until nothing changes anywhere."""
        comm, dev = self.comm, self.device
        eps32 = float(np.float32(eps))
        m = len(self.points)
        phase = {"setup": 0.0, "cluster": 0.0, "propagate": 0.0, "number": 0.0, "assign": 0.0}
        profile = self.profile

        def lap(name, t0):  # (TKNN_SHARD_PROFILE / bench.py's instrumented steps: a device synchronisation per phase)
            if profile:
                if dev.type == "cuda":
                    torch.cuda.synchronize(dev)
                phase[name] += (time.perf_counter() - t0) * 1e3
            return time.perf_counter()

        t = time.perf_counter()
        halo, counts_in, sent_local, ids = self._db_setup(eps32, reuse=_reuse_setup)
        t = lap("setup", t)
        local = self.db_engine.dbscan(eps32, min_pts)
        t = lap("cluster", t)
        core = local["core"].to(dev).bool()
        big = torch.iinfo(torch.int64).max
        # Label propagation works on CLUSTERS, not points (round 4; the per-point form spent 160 ms of a 174 ms step in
        # scatter-reduces of ten million values onto 83 addresses): lab_comp[c] = the smallest global core id known for local
        # cluster c.  Per point there is one step before the rounds (the clusters' own minima: tknnSegmentMin) and one gather
        # after them; a round moves the values of the halo rows and touches the clusters they belong to.
        seg = torch.where(core, local["labels"].to(dev).int(), torch.full((len(ids),), -1, dtype=torch.int32, device=dev))  # cluster of a core point, else -1
        ncomp = int((local.get("info") or {}).get("clusters", -1))
        if ncomp < 0:  # (an engine that does not say)
            ncomp = int(seg.max().item()) + 1 if len(seg) else 0
        ncomp = max(ncomp, 0)
        lab_comp = torch.full((max(ncomp, 1),), big, dtype=torch.int64, device=dev)
        self._seg_min(seg, ids.long(), lab_comp)
        seg_halo = seg[m:]
        seg_sent = [seg[:m][sent_local[p]] if len(sent_local[p]) else seg[:0] for p in range(comm.world)]
        offs = np.concatenate([[0], np.cumsum(counts_in)])
        big_row = torch.full((1,), big, dtype=torch.int64, device=dev)

        def of_clusters(which):  # the clusters' labels for a list of rows (a row that is not core: nothing to say)
            return torch.where(which >= 0, lab_comp[which.clamp(min=0).long()], big_row).reshape(-1, 1).contiguous()

        rounds = 0
        while True:
            rounds += 1
            before = lab_comp.clone()
            # copies -> owners (minimum)
            back = [of_clusters(seg_halo[int(offs[s_]): int(offs[s_ + 1])]) for s_ in range(comm.world)]
            # (both directions answer rows that travelled before: every count is known at both ends, none is exchanged)
            expect_back = [int(len(sent_local[p])) for p in range(comm.world)]
            expect_back[comm.rank] = int(len(back[comm.rank]))
            for p, vals in enumerate(comm.exchange_rows(back, 1, torch.int64, dev, counts_in=expect_back)):
                if p != comm.rank and len(vals):
                    self._seg_min(seg_sent[p], vals.flatten(), lab_comp)
            # owners -> copies
            fwd = [of_clusters(seg_sent[p]) for p in range(comm.world)]
            expect_fwd = [int(c) for c in counts_in]
            expect_fwd[comm.rank] = int(len(fwd[comm.rank]))
            new = comm.exchange_rows(fwd, 1, torch.int64, dev, counts_in=expect_fwd)
            new[comm.rank] = new[comm.rank][:0]
            if len(halo):
                self._seg_min(seg_halo, torch.cat(new, dim=0).flatten(), lab_comp)
            changed = torch.tensor([int(bool((lab_comp != before).any()))], dtype=torch.int64, device=dev)
            comm.all_reduce(changed, dist.ReduceOp.MAX)
            if int(changed.item()) == 0:
                break
        t = lap("propagate", t)
        # cluster numbers: ascending smallest core id over ALL ranks (of the clusters that have a core point of MINE)
        own_min = torch.full_like(lab_comp, big)
        self._seg_min(seg[:m], ids[:m].long(), own_min)
        mine = torch.unique(lab_comp[own_min != big])
        width = torch.tensor([len(mine)], dtype=torch.int64, device=dev)
        comm.all_reduce(width, dist.ReduceOp.MAX)
        padded = torch.full((max(int(width.item()), 1),), big, dtype=torch.int64, device=dev)
        padded[: len(mine)] = mine
        everyone = torch.unique(comm.all_gather(padded).flatten())
        everyone = everyone[everyone != big]
        number_comp = torch.searchsorted(everyone, lab_comp.clamp(max=int(everyone[-1].item()) if len(everyone) else 0)).int()
        core_label = torch.where(seg >= 0, number_comp[seg.clamp(min=0).long()], torch.full_like(seg, -1))
        t = lap("number", t)
        labels = self.db_engine.dbscan_assign(eps32, core_label)
        t = lap("assign", t)
        info = {"clusters": int(len(everyone)), "rounds": rounds, "label_rounds": rounds, "halo_points": int(len(halo)), "halo_exchanges": 1}
        # the tile's own clustering as its engine reports it (device times of the three traversal kernels, work counters)
        eng_info = dict(local.get("info") or {})
        eng_info["n_clustered"] = int(len(ids))
        info["engine"] = eng_info
        if profile:
            info["phase_ms"] = phase
        return {"labels": labels[:m].to(dev), "core": core[:m], "info": info}

    def _seg_min(self, seg, val, out):
        """out[seg[i]] = min(out[seg[i]], val[i]) for seg[i] >= 0: the engine's kernel (tknnSegmentMin) if it has one, else torch"""
        if len(seg) == 0:
            return out
        if hasattr(self.db_engine, "segment_min"):
            return self.db_engine.segment_min(seg.int(), val, out)
        keep = seg >= 0
        if bool(keep.any()):
            out.scatter_reduce_(0, seg[keep].long(), val[keep], "amin")
        return out

    def _db_setup(self, eps32, reuse=False):
        """The engine over my tile plus a halo of radius 2 eps (see dbscan).  ``reuse``: the auto-eps loop's last growth
        round and the clustering that follows it use one exchange and one tree; every other call does its own."""
        if reuse and self._db_cache is not None and self._db_cache[0] == eps32:
            return self._db_cache[1]
        comm, dev = self.comm, self.device
        blocks = self._halo_blocks(2.0 * eps32)
        got = comm.exchange_rows(blocks, 4, torch.float32, dev)
        got[comm.rank] = got[comm.rank][:0]
        counts_in = [len(g) for g in got]
        halo = torch.cat(got, dim=0)
        # the rows I sent, as positions in my own arrays (values travel in the same order later)
        id_sorted, id_perm = torch.sort(self.ids.long())
        sent_local = [id_perm[torch.searchsorted(id_sorted, b[:, 3].contiguous().view(torch.int32).long())]
                      if len(b) else torch.zeros(0, dtype=torch.long, device=dev) for b in blocks]
        pts = torch.cat([self.points, halo[:, :3]], dim=0).contiguous()
        ids = torch.cat([self.ids, halo[:, 3].contiguous().view(torch.int32)], dim=0).contiguous()
        if self.db_engine is None:
            self.db_engine = self._engine_factory(dev)
        self.db_engine.build(pts, ids)
        self._db_cache = (eps32, (halo, counts_in, sent_local, ids))
        return self._db_cache[1]

    def dbscan_auto(self, eps0, min_pts, max_noise=0.05, max_rounds=32):
        """RT-DBSCAN with an auto-grown eps over the tiles (spec: oracle/dbscan_oracle.c, dbref_dbscan_auto): rounds of the
        eps doubling (fp32, hostCode.cpp:321) until at most floor(max_noise * n_total) points of the WHOLE set are noise.
        A growth round exchanges the 2-eps halo, builds the tile's tree and counts its own noise points
        (tknnDbscanNoise) -- one 8-byte all-reduce per round; clusters are built once, at the final eps."""
        comm, dev = self.comm, self.device
        bound = int(np.floor(float(max_noise) * self.n_total))
        eps = np.float32(eps0)
        m = len(self.points)
        for t in range(int(max_rounds)):
            # a growth round only counts: core flags are exact for my own points inside the 2-eps halo, and a point of
            # mine is noise iff it is not core and has no core point within eps -- no clusters, no label exchange
            self._db_setup(float(eps))
            flags = self.db_engine.dbscan_noise(float(eps), min_pts)["noise"]
            noise = torch.tensor([int(flags[:m].sum().item())], dtype=torch.int64, device=dev)
            comm.all_reduce(noise, dist.ReduceOp.SUM)
            if int(noise.item()) <= bound:
                r = self.dbscan(float(eps), min_pts, _reuse_setup=True)  # (same eps: the round's halo and tree)
                # "rounds" = growth rounds, as in tknnDbscanAutoInfo; the label propagation's rounds stay in "label_rounds"
                r["info"].update({"rounds": t + 1, "eps_rounds": t + 1, "eps": float(eps), "noise": int(noise.item())})
                return r
            eps = np.float32(eps * np.float32(2))
        raise _lib.TknnError(-4, "max_rounds doublings of eps did not bring the noise under the bound")

    def gather_rows(self):
        """(global_ids, idx, dist, intersections) of every rank concatenated on every rank (tests)."""
        comm, dev = self.comm, self.device
        k = self.last["idx"].shape[1]
        rows = torch.cat([self.ids.view(-1, 1).to(torch.float64), self.last["idx"].to(torch.float64),
                          self.last["dist"].to(torch.float64),
                          self.last["intersections"].view(-1, 1).to(torch.float64)], dim=1)
        blocks = [rows for _ in range(comm.world)]
        got = torch.cat(comm.exchange_rows(blocks, rows.shape[1], torch.float64, dev), dim=0).cpu().numpy()
        order = np.argsort(got[:, 0])
        got = got[order]
        return (got[:, 0].astype(np.int64), got[:, 1:1 + k].astype(np.int32),
                got[:, 1 + k:1 + 2 * k].astype(np.float32), got[:, 1 + 2 * k].astype(np.int64))
