"""Host side of the TrueKNN engine above the C-ABI (include/owlknn.h).

Mirrors what ``samples/s01-trueknn/hostCode.cpp`` does with its command line
``file n dim start_radius k timefile``: upload points, build the accel (hostCode.cpp:201-206),
run the radius-doubling solve (hostCode.cpp:285-340) and expose the frameBuffer rows.
torch is used only to own device memory and the stream.
"""
import ctypes

import numpy as np

from . import _lib
from .datasets import pad_to_3d

NEIGH_BYTES = 24  # GeomTypes.h:22-28


class TrueKNN:
    """One engine on one GPU.  ``points`` may be a numpy array (n,2|3) or a CUDA float32 tensor (n,3)."""

    def __init__(self, device=None):
        import torch

        self._torch = torch
        lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("TrueKNN needs an MI355X: no GPU is visible and there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self._lib = lib
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.tknnCreate(ctypes.byref(self._h)))
        self.n = 0
        self.build_info = None
        self.last_info = None

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.tknnDestroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return ctypes.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def _points(self, points):
        torch = self._torch
        if isinstance(points, np.ndarray):
            points = torch.from_numpy(pad_to_3d(points)).to(self.device)
        if points.dtype != torch.float32 or points.dim() != 2 or points.shape[1] != 3 or not points.is_cuda:
            raise ValueError("points must be float32 (n,3) on the GPU")
        return points.contiguous()

    def _ids(self, ids, n):
        torch = self._torch
        if ids is None:
            return None
        if isinstance(ids, np.ndarray):
            ids = torch.from_numpy(np.ascontiguousarray(ids, dtype=np.int32)).to(self.device)
        if ids.dtype != torch.int32 or ids.dim() != 1 or ids.shape[0] != n or not ids.is_cuda:
            raise ValueError("ids must be int32 (n,) on the GPU")
        return ids.contiguous()

    def build(self, points, ids=None):
        """LBVH over the points.  ``ids`` (optional int32) are the identities reported in neighbour
        lists (global indices when these points are one tile of a larger set)."""
        torch = self._torch
        points = self._points(points)
        ids = self._ids(ids, points.shape[0])
        info = _lib.BuildInfo()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.tknnBuildIds(self._h, ctypes.c_void_p(points.data_ptr()),
                                              None if ids is None else ctypes.c_void_p(ids.data_ptr()),
                                              points.shape[0], ctypes.byref(info), self._stream()))
        self.n = int(points.shape[0])
        self.build_info = info.as_dict()
        return self.build_info

    def set_halo(self, points=None, ids=None):
        """Second point set every query also searches (border points of neighbouring tiles)."""
        torch = self._torch
        with torch.cuda.device(self.device):
            if points is None or len(points) == 0:
                _lib.check(self._lib.tknnSetHalo(self._h, None, None, 0, self._stream()))
                return
            points = self._points(points)
            ids = self._ids(ids, points.shape[0])
            if ids is None:
                raise ValueError("halo points need ids")
            _lib.check(self._lib.tknnSetHalo(self._h, ctypes.c_void_p(points.data_ptr()),
                                             ctypes.c_void_p(ids.data_ptr()), points.shape[0], self._stream()))
            torch.cuda.current_stream(self.device).synchronize()  # the engine copied what it needs

    def halo_select(self, boxes, box_peer, npeers):
        """Send side of the halo exchange: my points inside any of ``boxes`` (m,6 float32 closed boxes,
        lo xyz hi xyz) of each peer ``box_peer[j]``.  Returns (rows, counts): rows (sum(counts), 4)
        float32 wire rows x y z id-bits with peer p's rows contiguous (in peer order), counts a
        python list of length npeers.  Row order inside a peer's segment is unspecified."""
        torch = self._torch
        with torch.cuda.device(self.device):
            dev = self.device
            boxes = torch.as_tensor(boxes, dtype=torch.float32, device=dev).contiguous().view(-1, 6)
            box_peer = torch.as_tensor(box_peer, dtype=torch.int32, device=dev).contiguous()
            counts = torch.empty(npeers, dtype=torch.int64, device=dev)
            args = (self._h, ctypes.c_void_p(boxes.data_ptr()), ctypes.c_void_p(box_peer.data_ptr()), int(len(box_peer)), int(npeers))
            _lib.check(self._lib.tknnHaloSelect(*args, ctypes.c_void_p(counts.data_ptr()), None, None, self._stream()))
            host_counts = counts.cpu()  # the caller needs them on the host anyway (message sizes)
            offsets = (torch.cumsum(host_counts, 0) - host_counts).to(dev)
            total = int(host_counts.sum())
            rows = torch.empty((total, 4), dtype=torch.float32, device=dev)
            if total:
                _lib.check(self._lib.tknnHaloSelect(*args, None, ctypes.c_void_p(offsets.data_ptr()),
                                                    ctypes.c_void_p(rows.data_ptr()), self._stream()))
            return rows, host_counts.tolist()

    def halo_select_fixed(self, boxes, box_peer, npeers, caps):
        """The one-pass form of ``halo_select`` for the fixed-capacity exchange (tknnHaloSelectFixed): ``caps[p]`` rows per peer.
        Returns (messages, starts, counts): ``messages`` a (sum(caps) + npeers, 4) float32 buffer in which peer p's message is
        rows starts[p] .. starts[p] + caps[p] -- a header row (the count, as two float32 cells that hold integers below 2**24
        exactly) and caps[p] wire rows, those past the count unspecified --, ``counts`` the exact per-peer counts as an int64
        DEVICE tensor (a count above its capacity: rows were dropped, the caller falls back to ``halo_select``).  Nothing in
        here waits for the device."""
        torch = self._torch
        with torch.cuda.device(self.device):
            dev = self.device
            boxes = torch.as_tensor(boxes, dtype=torch.float32, device=dev).contiguous().view(-1, 6)
            box_peer = torch.as_tensor(box_peer, dtype=torch.int32, device=dev).contiguous()
            caps = [int(c) for c in caps]
            starts, at = [], 0
            for c in caps:
                starts.append(at)
                at += c + 1
            messages = torch.empty((at, 4), dtype=torch.float32, device=dev)
            caps_dev = torch.tensor(caps, dtype=torch.int64, device=dev)
            offsets = torch.tensor([s0 + 1 for s0 in starts], dtype=torch.int64, device=dev)
            counts = torch.empty(npeers, dtype=torch.int64, device=dev)
            _lib.check(self._lib.tknnHaloSelectFixed(self._h, ctypes.c_void_p(boxes.data_ptr()), ctypes.c_void_p(box_peer.data_ptr()), int(len(box_peer)),
                                                     int(npeers), ctypes.c_void_p(caps_dev.data_ptr()), ctypes.c_void_p(offsets.data_ptr()),
                                                     ctypes.c_void_p(messages.data_ptr()), ctypes.c_void_p(counts.data_ptr()), self._stream()))
            heads = torch.tensor(starts, dtype=torch.int64, device=dev)
            messages[heads, 0] = (counts % (1 << 24)).float()
            messages[heads, 1] = (counts >> 24).float()
            return messages, starts, counts

    supports_phases = True  # solve(phase=1 | 2): interior / boundary queries of a tile (tknnSolveOptions.phase)

    def solve(self, k, start_radius, kernel=_lib.KERNEL_AUTO, max_rounds=64, want_fb=False,
              out=None, want_levels=False, allow_unfinished=False, phase=0, stream=None, start_radii=None, fb_only=False):
        """Returns dict(idx (n,k) int32, dist (n,k) f32, intersections (n,) int64[, fb (n*k*24,) uint8])
        as CUDA tensors plus ``info``.  ``out`` may carry preallocated tensors of those names.
        ``phase`` 1 / 2: only the interior / boundary queries marked by the last ``halo_select`` (sharded use);
        ``stream``: a torch.cuda.Stream to launch on instead of the current one (a phase-1 solve runs on a
        stream and host thread of its own beside the halo exchange).
        ``start_radii``: (n,) float32, a start radius per query (row) instead of ``start_radius`` for all -- the opt-in
        per-query radius schedule (tknnSolveOptions.d_start_radii).
        ``fb_only``: only the reference's frameBuffer records are written (d_idx = d_dist = d_intersections = NULL), the
        way the reference's own host code receives its results."""
        torch = self._torch
        n = self.n
        out = dict(out or {})
        with torch.cuda.device(self.device):
            if n > 0 and k > 0:
                if not fb_only:
                    out.setdefault("idx", torch.empty((n, k), dtype=torch.int32, device=self.device))
                    out.setdefault("dist", torch.empty((n, k), dtype=torch.float32, device=self.device))
                    out.setdefault("intersections", torch.empty((n,), dtype=torch.int64, device=self.device))
                if want_fb or fb_only:
                    out.setdefault("fb", torch.empty((n * k * NEIGH_BYTES,), dtype=torch.uint8, device=self.device))
                if want_levels or allow_unfinished:
                    out.setdefault("levels", torch.empty((n,), dtype=torch.int32, device=self.device))
            info = _lib.SolveInfo()

            def ptr(name):
                t = out.get(name)
                return ctypes.c_void_p(t.data_ptr()) if t is not None else None

            opt = _lib.SolveOptions()
            opt.k, opt.start_radius, opt.kernel = int(k), float(start_radius), int(kernel)
            opt.max_rounds, opt.allow_unfinished = int(max_rounds), int(bool(allow_unfinished))
            opt.phase = int(phase)
            radii = None
            if start_radii is not None:
                radii = torch.as_tensor(start_radii, dtype=torch.float32, device=self.device).contiguous()
                if radii.shape != (n,):
                    raise ValueError("start_radii must have one entry per point")
                opt.d_start_radii = radii.data_ptr()
            for field, name in (("d_idx", "idx"), ("d_dist", "dist"), ("d_intersections", "intersections"),
                                ("d_fb", "fb"), ("d_levels", "levels")):
                p = ptr(name)
                setattr(opt, field, p.value if p is not None else None)
            launch_on = self._stream() if stream is None else ctypes.c_void_p(stream.cuda_stream)
            _lib.check(self._lib.tknnSolveEx(self._h, ctypes.byref(opt), ctypes.byref(info), launch_on))
        self.last_info = info.as_dict()
        out["info"] = self.last_info
        return out

    def repair_exact(self, result, k, start_radius):
        """Turn the rows of ``solve(..., want_levels=True)`` into exact kNN in place (opt-in; the
        reference's box-candidate rows are what ``solve`` returns).  Returns the number of rows rewritten."""
        torch = self._torch
        n_fixed = ctypes.c_int64(0)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.tknnRepairExact(
                self._h, int(k), ctypes.c_float(start_radius), ctypes.c_void_p(result["levels"].data_ptr()),
                ctypes.c_void_p(result["idx"].data_ptr()), ctypes.c_void_p(result["dist"].data_ptr()),
                ctypes.byref(n_fixed), self._stream()))
        return int(n_fixed.value)

    def dbscan(self, eps, min_pts, want_counts=False):
        """RT-DBSCAN over the built tree: dict(labels int32 (n,), core bool (n,), [counts], info)."""
        torch = self._torch
        n = self.n
        with torch.cuda.device(self.device):
            labels = torch.empty((n,), dtype=torch.int32, device=self.device)
            core = torch.empty((n,), dtype=torch.uint8, device=self.device)
            counts = torch.empty((n,), dtype=torch.int32, device=self.device) if want_counts else None
            info = _lib.DbscanInfo()
            _lib.check(self._lib.tknnDbscan(self._h, ctypes.c_float(eps), int(min_pts), ctypes.c_void_p(labels.data_ptr()),
                                            ctypes.c_void_p(core.data_ptr()),
                                            None if counts is None else ctypes.c_void_p(counts.data_ptr()),
                                            ctypes.byref(info), self._stream()))
        out = {"labels": labels, "core": core.view(torch.bool), "info": info.as_dict()}
        if counts is not None:
            out["counts"] = counts
        return out

    def dbscan_auto(self, eps0, min_pts, max_noise=0.05, max_rounds=32):
        """RT-DBSCAN with an auto-grown eps (tknnDbscanAuto): eps doubles from ``eps0`` until at most
        floor(max_noise * n) points are noise.  dict(labels, core, info(rounds, eps, noise, clusters, ...))."""
        torch = self._torch
        n = self.n
        with torch.cuda.device(self.device):
            labels = torch.empty((n,), dtype=torch.int32, device=self.device)
            core = torch.empty((n,), dtype=torch.uint8, device=self.device)
            info = _lib.DbscanAutoInfo()
            _lib.check(self._lib.tknnDbscanAuto(self._h, ctypes.c_float(eps0), int(min_pts), ctypes.c_double(max_noise), int(max_rounds),
                                                ctypes.c_void_p(labels.data_ptr()), ctypes.c_void_p(core.data_ptr()),
                                                ctypes.byref(info), self._stream()))
        return {"labels": labels, "core": core.view(torch.bool), "info": info.as_dict()}

    def dbscan_noise(self, eps, min_pts):
        """Which points tknnDbscan(eps, min_pts) would label -1, without building clusters (tknnDbscanNoise: one growth
        round of the auto-eps loop for a caller that runs the loop itself).  dict(noise (n,) bool, count)."""
        torch = self._torch
        with torch.cuda.device(self.device):
            noise = torch.empty((self.n,), dtype=torch.uint8, device=self.device)
            count = ctypes.c_int64(0)
            _lib.check(self._lib.tknnDbscanNoise(self._h, ctypes.c_float(eps), int(min_pts), ctypes.c_void_p(noise.data_ptr()),
                                                 ctypes.byref(count), self._stream()))
        return {"noise": noise.view(torch.bool), "count": int(count.value)}

    def dbscan_assign(self, eps, core_label):
        """Last step of DBSCAN with labels decided by the caller: ``core_label`` (n,) int32, >= 0 for
        core points.  Returns labels (n,) int32: core points keep theirs, the others take the smallest
        label among the core points within eps, -1 if there is none."""
        torch = self._torch
        with torch.cuda.device(self.device):
            core_label = torch.as_tensor(core_label, dtype=torch.int32, device=self.device).contiguous()
            if core_label.shape != (self.n,):
                raise ValueError("core_label must have one entry per point")
            labels = torch.empty((self.n,), dtype=torch.int32, device=self.device)
            info = _lib.DbscanInfo()
            _lib.check(self._lib.tknnDbscanAssign(self._h, ctypes.c_float(eps), ctypes.c_void_p(core_label.data_ptr()),
                                                  ctypes.c_void_p(labels.data_ptr()), ctypes.byref(info), self._stream()))
        return labels

    def segment_min(self, segment, value, out):
        """out[segment[i]] = min(out[segment[i]], value[i]) for segment[i] >= 0, in place (tknnSegmentMin): ``segment`` (n,)
        int32, ``value`` (n,) int64, ``out`` (m,) int64 preset by the caller, all on the engine's device."""
        torch = self._torch
        if segment.dtype != torch.int32 or value.dtype != torch.int64 or out.dtype != torch.int64 or segment.shape != value.shape:
            raise ValueError("segment_min: int32 segments, int64 values of the same shape, int64 out")
        with torch.cuda.device(self.device):
            segment, value = segment.contiguous(), value.contiguous()
            _lib.check(self._lib.tknnSegmentMin(self._h, ctypes.c_void_p(segment.data_ptr()), ctypes.c_void_p(value.data_ptr()), int(segment.numel()),
                                                ctypes.c_void_p(out.data_ptr()), self._stream()))
        return out

    def export_tree(self):
        """Host copies of the LBVH for tests: nodes (n-1,8) uint32 view, ropes, prim ids."""
        torch = self._torch
        n = self.n
        nodes = np.zeros((max(n - 1, 1), 8), np.uint32)
        rope_node = np.zeros(max(n - 1, 1), np.int32)
        rope_leaf = np.zeros(n, np.int32)
        prim = np.zeros(n, np.int32)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.tknnExportTree(self._h, nodes.ctypes.data, rope_node.ctypes.data,
                                                rope_leaf.ctypes.data, prim.ctypes.data, self._stream()))
        split_owner = np.zeros(max(n - 1, 1), np.int32)
        paths = np.zeros(((n + 63) // 64, 5), np.int32)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.tknnExportTreeTables(self._h, split_owner.ctypes.data, paths.ctypes.data, self._stream()))
        return {"nodes": nodes[: max(n - 1, 0)], "rope_node": rope_node[: max(n - 1, 0)],
                "rope_leaf": rope_leaf, "prim_id": prim, "split_owner": split_owner[: max(n - 1, 0)], "block_paths": paths}


def trueknn(points, k, start_radius, **kw):
    """One-shot helper: build + solve, results as numpy arrays."""
    eng = TrueKNN()
    try:
        eng.build(points)
        r = eng.solve(k, start_radius, **kw)
        res = {name: (v.cpu().numpy() if hasattr(v, "cpu") else v) for name, v in r.items()}
        res["build_info"] = eng.build_info
        return res
    finally:
        eng.close()
