#!/usr/bin/env python3
"""Copy one scripts/profile_gpu.sh run of bench.py into profiles/ (bench line, summary, kernel stats)
and refresh profiles/hbm_traffic.json from its FETCH_SIZE / WRITE_SIZE passes.

    python scripts/update_profiles.py gpurun_out/bench_r01d.json gpurun_out/prof_bench_r01d
"""
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    bench, prof = sys.argv[1], sys.argv[2]
    shutil.copy(bench, os.path.join(ROOT, "profiles", "r01_bench.json"))
    shutil.copy(os.path.join(prof, "summary.txt"), os.path.join(ROOT, "profiles", "r01_bench_team_summary.txt"))
    stats = glob.glob(os.path.join(prof, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(ROOT, "profiles", "r01_bench_team_kernel_stats.csv"))
    text = open(os.path.join(prof, "summary.txt")).read()
    sec = text[text.index("== counters: void owlmi::(anonymous namespace)::team_kernel"):]
    sec = sec[: sec.index("== counters:", 10)] if "== counters:" in sec[10:] else sec
    val = lambda name: float(re.search(name + r"\s+total=\S+\s+launches=\d+\s+per_launch=(\S+)", sec).group(1))  # noqa: E731
    fetch_kb, write_kb = val("FETCH_SIZE"), val("WRITE_SIZE")
    p = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    d = json.load(open(p))
    key = "team_kernel:n=10000000:k=10"
    d[key]["bytes_per_launch"] = int((2 * fetch_kb + write_kb) * 1024)
    d[key]["FETCH_SIZE_KB_per_launch"] = fetch_kb
    d[key]["WRITE_SIZE_KB_per_launch"] = write_kb
    d[key]["uncorrected_bytes_per_launch"] = int((fetch_kb + write_kb) * 1024)
    d[key]["valu_wave_instructions_per_launch"] = val("SQ_INSTS_VALU")
    d[key]["salu_wave_instructions_per_launch"] = val("SQ_INSTS_SALU")
    d[key]["gui_active_cycles_per_launch"] = val("GRBM_GUI_ACTIVE")  # summed over the 8 XCDs
    json.dump(d, open(p, "w"), indent=1)
    b = json.load(open(bench))
    print("bench: %.3g q/s, %.2f ms/step, kernel %.2f ms; VALU %.3g SALU %.3g per launch; traffic %.2f GB" % (
        b["value"], b["ms_per_step"], b["roofline"]["kernel_ms"], val("SQ_INSTS_VALU"), val("SQ_INSTS_SALU"),
        d[key]["bytes_per_launch"] / 1e9))


if __name__ == "__main__":
    main()
