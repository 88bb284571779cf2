#!/bin/bash
# On the GPU box: a few PMC passes over one command, one counter group per pass (never with other trace domains).
#   scripts/pmc_probe.sh <tag> "<group 1>" "<group 2>" ... -- python3 scripts/quick_bench.py 10000000 10 3 2
# Prints, per kernel matching $PMC_KERNEL (default team_kernel), counter totals divided by launches.
tag=$1; shift
groups=()
while [ "$1" != "--" ]; do groups+=("$1"); shift; done
shift
out=$PWD/gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
i=0
for g in "${groups[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $g --output-format csv -d $out/p$i -- "$@" > $out/p$i.log 2>&1 || { echo "pass $i ($g) failed"; tail -3 $out/p$i.log; }
done
python3 - "$out" "${PMC_KERNEL:-team_kernel}" <<'PY'
import csv, glob, sys, collections
out, pat = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(float); launches = collections.defaultdict(set)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if pat not in row.get("Kernel_Name", ""):
            continue
        c = row["Counter_Name"]; tot[c] += float(row["Counter_Value"]); launches[c].add((f, row.get("Dispatch_Id")))
for c in sorted(tot):
    print("%-34s per_launch=%.6g  (launches=%d)" % (c, tot[c] / max(len(launches[c]), 1), len(launches[c])))
PY
