#!/bin/bash
# GPU box: instruction-cache counters of the kernels of a tknnDbscan call (scripts/db_times.py, config 3)
out=$PWD/gpurun_out/prof_icache; rm -rf $out; mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/p1 -- python3 $root/scripts/db_times.py 10000000 3 > $out/p1.log 2>&1 || tail -5 $out/p1.log
rocprofv3 --kernel-trace --pmc SQ_IFETCH_LEVEL SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $out/p2 -- python3 $root/scripts/db_times.py 10000000 3 > $out/p2.log 2>&1 || tail -5 $out/p2.log
cd $root
python3 - <<PY
import csv, glob, collections
for p in ("p1", "p2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob("gpurun_out/prof_icache/%s/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].split("::")[-1]
            if not k.startswith("db_") and "team" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"] or 0)
    for k, d in acc.items():
        print(p, k, {c: "%.4g" % v for c, v in sorted(d.items())})
PY
