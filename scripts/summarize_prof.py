#!/usr/bin/env python3
"""Condense a scripts/profile_gpu.sh output directory into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    for path in glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True):
        print("== kernel stats (%s)" % os.path.relpath(path, root))
        with open(path) as fh:
            for i, row in enumerate(csv.reader(fh)):
                if i < 12:
                    print("  " + ", ".join(row))
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(int))
    each = defaultdict(lambda: defaultdict(list))  # the memory-side counters dispatch by dispatch (update_profiles.py selects from these)
    for path in sorted(glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "?")
                c = row.get("Counter_Name", "?")
                agg[k][c] += float(row.get("Counter_Value", 0) or 0)
                calls[k][c] += 1
                if c in ("FETCH_SIZE", "WRITE_SIZE"):
                    each[k][c].append(float(row.get("Counter_Value", 0) or 0))
    for k in sorted(agg, key=lambda s: -sum(agg[s].values())):
        short = k if len(k) < 90 else k[:87] + "..."
        print("== counters: %s" % short)
        for c in sorted(agg[k]):
            n = calls[k][c]
            print("  %-24s total=%.6g  launches=%d  per_launch=%.6g" % (c, agg[k][c], n, agg[k][c] / max(n, 1)))
            if c in each[k] and len(each[k][c]) > 1:
                print("  %-24s by dispatch: %s%s" % ("", " ".join("%.6g" % v for v in each[k][c][:16]), " ..." if len(each[k][c]) > 16 else ""))


if __name__ == "__main__":
    main()
