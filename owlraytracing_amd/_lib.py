"""ctypes binding of libowl_mi355x.so (the C-ABI declared in include/owlknn.h).

There is no fallback: if the library has not been built, or no MI355X is visible when an engine is
created, the call raises.  Build with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C owlraytracing_amd/csrc``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OWL_MI355X_LIB") or os.path.join(_HERE, "libowl_mi355x.so")  # override: diagnostic builds

KERNEL_AUTO, KERNEL_LANE, KERNEL_WAVE, KERNEL_TEAM = 0, 1, 2, 3
MAX_K = 1024  # include/owlknn.h TKNN_MAX_K (k <= 64: register lists; above: the team walk with the lists in memory)


class TknnError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("tknn error %d: %s" % (code, message))
        self.code = code


class SolveInfo(ctypes.Structure):
    _fields_ = [
        ("rounds", ctypes.c_int32),
        ("final_radius", ctypes.c_float),
        ("total_intersections", ctypes.c_int64),
        ("node_tests", ctypes.c_int64),
        ("point_tests", ctypes.c_int64),
        ("total_active_rounds", ctypes.c_int64),
        ("solve_ms", ctypes.c_float),
        ("dominant_kernel_ms", ctypes.c_float),
        ("dominant_kernel_launches", ctypes.c_int32),
        ("kernel_used", ctypes.c_int32),
        ("list_capacity", ctypes.c_int32),
        ("unfinished", ctypes.c_int64),
        ("tie_rows", ctypes.c_int64),
        ("tie_rows_left", ctypes.c_int64),
        ("tie_ms", ctypes.c_float),
        ("reserved_", ctypes.c_int32),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class SolveOptions(ctypes.Structure):
    _fields_ = [
        ("k", ctypes.c_int32),
        ("start_radius", ctypes.c_float),
        ("kernel", ctypes.c_int32),
        ("max_rounds", ctypes.c_int32),
        ("allow_unfinished", ctypes.c_int32),
        ("phase", ctypes.c_int32),
        ("d_idx", ctypes.c_void_p),
        ("d_dist", ctypes.c_void_p),
        ("d_intersections", ctypes.c_void_p),
        ("d_fb", ctypes.c_void_p),
        ("d_levels", ctypes.c_void_p),
        ("d_start_radii", ctypes.c_void_p),
    ]


class DbscanInfo(ctypes.Structure):
    _fields_ = [("clusters", ctypes.c_int32), ("solve_ms", ctypes.c_float), ("core_ms", ctypes.c_float),
                ("union_ms", ctypes.c_float), ("label_ms", ctypes.c_float), ("union_launches", ctypes.c_int32),
                ("node_tests", ctypes.c_int64), ("point_tests", ctypes.c_int64), ("core_point_tests", ctypes.c_int64),
                ("union_point_tests", ctypes.c_int64), ("label_point_tests", ctypes.c_int64),
                ("union_node_tests", ctypes.c_int64), ("groups", ctypes.c_int64)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_ if name != "pad_"}


class DbscanAutoInfo(ctypes.Structure):
    _fields_ = [("last", DbscanInfo), ("rounds", ctypes.c_int32), ("eps", ctypes.c_float), ("noise", ctypes.c_int64),
                ("probe_ms", ctypes.c_float), ("pad_", ctypes.c_int32)]

    def as_dict(self):
        d = {"rounds": self.rounds, "eps": self.eps, "noise": self.noise, "probe_ms": self.probe_ms}
        d.update(self.last.as_dict())
        return d


class BuildInfo(ctypes.Structure):
    _fields_ = [("build_ms", ctypes.c_float), ("device_bytes", ctypes.c_int64), ("n", ctypes.c_int32)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


# every symbol include/owlknn.h declares, with its signature
SIGNATURES = {
    "tknnLastError": (ctypes.c_char_p, []),
    "tknnDeviceCount": (ctypes.c_int, []),
    "tknnCreate": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p)]),
    "tknnDestroy": (None, [ctypes.c_void_p]),
    "tknnBuild": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                 ctypes.POINTER(BuildInfo), ctypes.c_void_p]),
    "tknnSolve": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int,
                                 ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                 ctypes.c_void_p, ctypes.POINTER(SolveInfo), ctypes.c_void_p]),
    "tknnBuildIds": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                    ctypes.POINTER(BuildInfo), ctypes.c_void_p]),
    "tknnSetHalo": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                   ctypes.c_void_p]),
    "tknnSolveEx": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(SolveOptions), ctypes.POINTER(SolveInfo),
                                   ctypes.c_void_p]),
    "tknnHaloSelect": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "tknnRepairExact": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.c_void_p]),
    "tknnDbscan": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                  ctypes.c_void_p, ctypes.POINTER(DbscanInfo), ctypes.c_void_p]),
    "tknnDbscanAssign": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.POINTER(DbscanInfo), ctypes.c_void_p]),
    "tknnDbscanAuto": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.POINTER(DbscanAutoInfo), ctypes.c_void_p]),
    "tknnHaloSelectFixed": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "tknnSegmentMin": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "tknnDbscanNoise": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64),
                                       ctypes.c_void_p]),
    "tknnExportTree": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "tknnExportTreeTables": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "tknnDebugThresholds": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
}

_lib = None


def load():
    """Load the shared library (no GPU needed for loading itself)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s not found: build the HIP extension first (make -C owlraytracing_amd/csrc); "
                "there is no CPU fallback" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        raise TknnError(rc, (load().tknnLastError() or b"").decode())


# native sources a kernel's code does NOT depend on, by kernel-name prefix: a committed profile of the packet kernel stays
# valid when only the clustering kernels change, and the other way round
_NOT_IN = {
    "team_": ("dbscan.hip", "halo_select.hip", "owl_runtime.cpp"),
    "db_": ("trueknn_team.hip", "trueknn_wave.hip", "halo_select.hip", "owl_runtime.cpp"),
}


def source_fingerprint(kernel=None):
    """16 hex digits naming the native sources the library -- or, with `kernel`, that kernel -- is built from (csrc/ and
    include/owlknn.h, include/owl/lbvh_device.h; without `kernel` also the other owl headers): what a committed rocprofv3 record must carry for bench.py
    to attach it to a run (profiles/hbm_traffic.json)."""
    import glob
    import hashlib

    root = os.path.dirname(_HERE)
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h"))
                   + glob.glob(os.path.join(_HERE, "csrc", "*.cpp")) + glob.glob(os.path.join(_HERE, "csrc", "Makefile"))
                   + glob.glob(os.path.join(root, "include", "**", "*.h"), recursive=True))
    if kernel is not None:
        skip = next((v for k, v in _NOT_IN.items() if kernel.startswith(k)), ())
        owl_headers = os.path.join(root, "include", "owl") + os.sep
        files = [f for f in files if os.path.basename(f) not in skip
                 and (not f.startswith(owl_headers) or os.path.basename(f) == "lbvh_device.h")]  # the tree's device layout
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.relpath(f, root).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
