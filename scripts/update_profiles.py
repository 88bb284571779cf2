#!/usr/bin/env python3
"""Copy one scripts/profile_gpu.sh run of bench.py into profiles/ (bench line, summary, kernel stats) and refresh
profiles/hbm_traffic.json -- the per-launch PMC figures bench.py attaches to its line -- from the run's passes.

    python scripts/update_profiles.py <tag> gpurun_out/bench_r03.json gpurun_out/prof_bench_r03 [--kernel NAME[:PERIOD]]...

Counters are taken PER DISPATCH from the passes' *_counter_collection.csv files, never as an average over a command:
the dispatches of a kernel are cut into calls of PERIOD launches (1 for team_kernel; 2 for db_group_union_kernel, whose
two passes per tknnDbscan call do different work), the first call is dropped (cold caches, lazy allocations), and what
is left must be ONE population -- position by position, the values may differ by a few per cent at most.  If they do
not (round 2: three of nine team_kernel launches of the profiled command also wrote the 2.4 GB frameBuffer, and the
"per launch" record was the mean of both kinds) the record is REFUSED: profile a command whose launches of that kernel
all do the benchmarked work (bench.py --no-fb-leg).

Without --kernel: the bench line's roofline kernel, and the three traversal kernels of its dbscan_config3 object.
The record carries the fingerprint of the native sources it was measured on (owlraytracing_amd._lib
.source_fingerprint); bench.py ignores a record whose fingerprint is not that of the sources it runs.
"""
import csv
import glob
import json
import os
import shutil
import statistics
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# work counters: a launch doing the benchmarked work reproduces them to a fraction of a per cent (the persistent kernels'
# work stealing moves a little); time-like counters (cycles, waits) depend on clocks and on what else ran
STRICT_TOL, LOOSE_TOL = 0.05, 0.35
LOOSE = ("GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
         "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC",
         "SQ_THREAD_CYCLES_VALU", "SQ_IFETCH", "TCC_HIT_sum", "TCC_MISS_sum")


class MixedDispatches(Exception):
    """the dispatches of a kernel in the profiled command are not one population"""


def dispatch_values(prof_dir, pattern):
    """{counter: [[value of dispatch 0, 1, ... of one profiled process], ...]} for the kernels whose name contains `pattern`,
    in dispatch order; rows of one dispatch and counter (one per instance of a multi-instance counter) are summed."""
    per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
    for path in sorted(glob.glob(os.path.join(prof_dir, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if pattern not in row.get("Kernel_Name", ""):
                    continue
                per[row["Counter_Name"]][path][int(row["Dispatch_Id"])] += float(row.get("Counter_Value", 0) or 0)
    return {c: [[v for _, v in sorted(d.items())] for _, d in sorted(files.items())] for c, files in per.items()}


def steady(runs, name, period=1, tol=None):
    """The per-launch figure of the benchmarked launches from the per-dispatch values of one counter (`runs`: one list per
    profiled process): each process's dispatches are cut into calls of `period` launches and its first call is dropped
    (cold); every position's values over all that is left must lie within `tol` (max / min - 1) of each other -- else
    MixedDispatches.  Returns (mean over the positions of the positions' medians, [median per position], launches used)."""
    if tol is None:
        tol = LOOSE_TOL if name in LOOSE else STRICT_TOL
    if runs and not isinstance(runs[0], (list, tuple)):
        runs = [runs]
    calls = []
    for values in runs:
        if period < 1 or len(values) % period:
            raise MixedDispatches("%s: %d dispatches are not whole calls of %d launches" % (name, len(values), period))
        calls += [values[i:i + period] for i in range(0, len(values), period)][1:]
    if not calls:
        raise MixedDispatches("%s: only one call of the kernel per profiled process (the first is dropped as cold)" % name)
    medians = []
    for pos in range(period):
        col = [c[pos] for c in calls]
        lo, hi = min(col), max(col)
        if hi > 0 and (lo <= 0 or hi / lo - 1.0 > tol):
            raise MixedDispatches(
                "%s differs by %.0f %% between launches of the profiled command (position %d of %d: %s): they do not all do the "
                "benchmarked work -- profile `bench.py --no-fb-leg`, or a command without other uses of the kernel" % (
                    name, 100.0 * (hi / lo - 1.0) if lo > 0 else float("inf"), pos, period, ", ".join("%.6g" % v for v in col[:12])))
        medians.append(statistics.median(col))
    return sum(medians) / period, medians, len(calls) * period


def build_record(prof_dir, pattern, period, fingerprint, command, tag):
    vals = dispatch_values(prof_dir, pattern)
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        raise SystemExit("no FETCH_SIZE / WRITE_SIZE dispatches of a kernel named *%s* under %s" % (pattern, prof_dir))
    got = {}
    for name, v in vals.items():
        got[name] = steady(v, name, period)

    def val(name):
        return got[name][0] if name in got else None

    fetch_kb, write_kb = val("FETCH_SIZE"), val("WRITE_SIZE")
    rec = {
        "source_sha16": fingerprint,
        "bytes_per_launch": int((2 * fetch_kb + write_kb) * 1024),
        "FETCH_SIZE_KB_per_launch": fetch_kb,
        "WRITE_SIZE_KB_per_launch": write_kb,
        "uncorrected_bytes_per_launch": int((fetch_kb + write_kb) * 1024),
        "launches_per_call": period,
        "dispatches_used": {"FETCH_SIZE": got["FETCH_SIZE"][2], "WRITE_SIZE": got["WRITE_SIZE"][2]},
        "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `%s` (profiles/%s_summary.txt), PER DISPATCH: "
               "calls of %d launch(es), the first call dropped, the rest required to agree within %d %% and reduced by their median; "
               "HBM-side bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: on gfx950 FETCH_SIZE reports half the bytes of 16-byte-per-lane reads "
               "(MI355X_MICROARCH.md, HBM section), the kernel's access width for points and boxes; WRITE_SIZE is taken as is" % (
                   command, tag, period, int(STRICT_TOL * 100)),
        "valu_wave_instructions_per_launch": val("SQ_INSTS_VALU"),
        "salu_wave_instructions_per_launch": val("SQ_INSTS_SALU"),
        "gui_active_cycles_per_launch": val("GRBM_GUI_ACTIVE"),
    }
    if period > 1:
        rec["per_position"] = {"FETCH_SIZE_KB": got["FETCH_SIZE"][1], "WRITE_SIZE_KB": got["WRITE_SIZE"][1]}
    if val("SQ_INSTS_BRANCH") is not None and val("SQ_INSTS_SALU") is not None:
        rec["scalar_pipe_instructions_per_launch"] = val("SQ_INSTS_SALU") + val("SQ_INSTS_BRANCH") + (val("SQ_INSTS_SMEM") or 0.0)
    if val("SQ_INSTS_LDS") is not None:
        rec["lds_instructions_per_launch"] = val("SQ_INSTS_LDS")
    wc = val("SQ_WAVE_CYCLES")
    if wc:
        for name, ctr in (("wave_wait_frac", "SQ_WAIT_ANY"), ("wave_issue_stall_frac", "SQ_WAIT_INST_ANY"), ("wave_active_frac", "SQ_ACTIVE_INST_ANY")):
            if val(ctr) is not None:
                rec[name] = val(ctr) / wc
    if val("TCC_HIT_sum") is not None and val("TCC_MISS_sum") is not None and val("TCC_HIT_sum") + val("TCC_MISS_sum") > 0:
        rec["l2_hit_frac"] = val("TCC_HIT_sum") / (val("TCC_HIT_sum") + val("TCC_MISS_sum"))
    return rec


def main():
    argv = sys.argv[1:]
    kernels = []
    while "--kernel" in argv:
        i = argv.index("--kernel")
        name, _, per = argv[i + 1].partition(":")
        kernels.append((name, int(per) if per else 1))
        del argv[i:i + 2]
    tag, bench, prof = argv[0], argv[1], argv[2]
    b = json.load(open(bench))
    from owlraytracing_amd import _lib

    cfg = b["config"]
    n_local = cfg["n_points_total"] // max(b.get("n_gpus", 1), 1)
    jobs = []  # (kernel, launches per call, key of the record)
    k_main = cfg.get("k", cfg.get("min_pts", 0))
    if kernels:
        jobs = [(name, per, "%s:n=%d:k=%d" % (name, n_local, k_main)) for name, per in kernels]
    else:
        r = b["roofline"]
        jobs.append((r["kernel"], int(r.get("launches_per_step", 1)), "%s:n=%d:k=%d" % (r["kernel"], n_local, k_main)))
        if "dbscan_config3" in b:  # the default run's RT-DBSCAN leg: BASELINE config 3 (10 M points, minPts 4)
            for name, rr in b["dbscan_config3"]["roofline"].items():
                jobs.append((name, int(rr.get("launches_per_step", 1)), "%s:n=%d:k=%d" % (name, 10_000_000, 4)))
            # (round 4: the label pass's second launch, whose traffic bench.py adds to db_label_kernel's)
            jobs.append(("db_rows_from_slots_kernel", 1, "db_rows_from_slots_kernel:n=%d:k=%d" % (10_000_000, 4)))
        for name, rr in r.get("kernels", {}).items():  # bench.py --workload dbscan
            if name != r["kernel"]:
                jobs.append((name, int(rr.get("launches_per_step", 1)), "%s:n=%d:k=%d" % (name, n_local, k_main)))
    shutil.copy(bench, os.path.join(ROOT, "profiles", "%s_bench.json" % tag))
    shutil.copy(os.path.join(prof, "summary.txt"), os.path.join(ROOT, "profiles", "%s_summary.txt" % tag))
    stats = glob.glob(os.path.join(prof, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(ROOT, "profiles", "%s_kernel_stats.csv" % tag))
    command = open(os.path.join(prof, "command.txt")).read().strip() if os.path.exists(os.path.join(prof, "command.txt")) else "bench.py"
    p = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    d = json.load(open(p)) if os.path.exists(p) else {}
    for kernel, period, key in jobs:
        try:
            rec = build_record(prof, kernel, period, _lib.source_fingerprint(kernel), command, tag)
        except MixedDispatches as e:
            raise SystemExit("REFUSED %s: %s" % (key, e))
        d[key] = rec
        print("%s: traffic %.3f GB per launch (FETCH %.0f KB x2 + WRITE %.0f KB), VALU %.3g SALU %.3g per launch, waits %.0f%% of wave cycles, %d dispatches" % (
            key, rec["bytes_per_launch"] / 1e9, rec["FETCH_SIZE_KB_per_launch"], rec["WRITE_SIZE_KB_per_launch"], rec["valu_wave_instructions_per_launch"] or 0,
            rec["salu_wave_instructions_per_launch"] or 0, 100 * rec.get("wave_wait_frac", 0), rec["dispatches_used"]["WRITE_SIZE"]))
    json.dump(d, open(p, "w"), indent=1)


if __name__ == "__main__":
    main()
