#!/usr/bin/env python3
"""Copy one scripts/profile_gpu.sh run of bench.py into profiles/ (bench line, summary, kernel stats) and refresh
profiles/hbm_traffic.json -- the per-launch PMC figures bench.py attaches to its line -- from the run's passes.

    python scripts/update_profiles.py <tag> gpurun_out/bench_r02.json gpurun_out/prof_bench_r02 [kernel-name-substring]

The record carries the fingerprint of the native sources it was measured on (owlraytracing_amd._lib
.source_fingerprint); bench.py ignores a record whose fingerprint is not that of the sources it runs.
"""
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    tag, bench, prof = sys.argv[1], sys.argv[2], sys.argv[3]
    b = json.load(open(bench))
    kernel = b["roofline"]["kernel"]
    pat = sys.argv[4] if len(sys.argv) > 4 else kernel
    shutil.copy(bench, os.path.join(ROOT, "profiles", "%s_bench.json" % tag))
    shutil.copy(os.path.join(prof, "summary.txt"), os.path.join(ROOT, "profiles", "%s_summary.txt" % tag))
    stats = glob.glob(os.path.join(prof, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(ROOT, "profiles", "%s_kernel_stats.csv" % tag))
    text = open(os.path.join(prof, "summary.txt")).read()
    secs = [s for s in text.split("== counters: ")[1:] if pat in s.split("\n", 1)[0]]
    if not secs:
        raise SystemExit("no counter section for a kernel named *%s* in %s/summary.txt" % (pat, prof))
    sec = secs[0]

    def val(name, default=None):
        m = re.search(r"^\s*" + name + r"\s+total=\S+\s+launches=\d+\s+per_launch=(\S+)", sec, re.M)
        return float(m.group(1)) if m else default

    from owlraytracing_amd import _lib

    fetch_kb, write_kb = val("FETCH_SIZE"), val("WRITE_SIZE")
    p = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    d = json.load(open(p)) if os.path.exists(p) else {}
    cfg = b["config"]
    n_local = cfg["n_points_total"] // max(b.get("n_gpus", 1), 1)
    key = "%s:n=%d:k=%d" % (kernel, n_local, cfg.get("k", cfg.get("min_pts", 0)))
    rec = {
        "source_sha16": _lib.source_fingerprint(kernel),
        "bytes_per_launch": int((2 * fetch_kb + write_kb) * 1024),
        "FETCH_SIZE_KB_per_launch": fetch_kb,
        "WRITE_SIZE_KB_per_launch": write_kb,
        "uncorrected_bytes_per_launch": int((fetch_kb + write_kb) * 1024),
        "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `%s` (profiles/%s_summary.txt); HBM-side bytes = "
               "(2*FETCH_SIZE + WRITE_SIZE)*1024: on gfx950 FETCH_SIZE reports half the bytes of 16-byte-per-lane reads "
               "(MI355X_MICROARCH.md, HBM section), the kernel's access width for points and boxes; WRITE_SIZE is taken as is" % (
                   open(os.path.join(prof, "command.txt")).read().strip() if os.path.exists(os.path.join(prof, "command.txt")) else "bench.py", tag),
        "valu_wave_instructions_per_launch": val("SQ_INSTS_VALU"),
        "salu_wave_instructions_per_launch": val("SQ_INSTS_SALU"),
        "gui_active_cycles_per_launch": val("GRBM_GUI_ACTIVE"),
    }
    if val("SQ_INSTS_BRANCH") is not None and val("SQ_INSTS_SALU") is not None:
        rec["scalar_pipe_instructions_per_launch"] = val("SQ_INSTS_SALU") + val("SQ_INSTS_BRANCH") + (val("SQ_INSTS_SMEM") or 0.0)
    if val("SQ_INSTS_LDS") is not None:
        rec["lds_instructions_per_launch"] = val("SQ_INSTS_LDS")
    wc = val("SQ_WAVE_CYCLES")
    if wc:
        for name, ctr in (("wave_wait_frac", "SQ_WAIT_ANY"), ("wave_issue_stall_frac", "SQ_WAIT_INST_ANY"), ("wave_active_frac", "SQ_ACTIVE_INST_ANY")):
            if val(ctr) is not None:
                rec[name] = val(ctr) / wc
    d[key] = rec
    json.dump(d, open(p, "w"), indent=1)
    print("%s: %.4g %s, %.2f ms/step, kernel %.2f ms; VALU %.3g SALU %.3g per launch; traffic %.2f GB; waits %.0f%% of wave cycles" % (
        key, b["value"], b["unit"], b["ms_per_step"], b["roofline"]["kernel_ms"], rec["valu_wave_instructions_per_launch"] or 0,
        rec["salu_wave_instructions_per_launch"] or 0, rec["bytes_per_launch"] / 1e9, 100 * rec.get("wave_wait_frac", 0)))


if __name__ == "__main__":
    main()
