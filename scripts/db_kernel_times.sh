#!/bin/bash
# GPU box: rocprofv3 kernel trace of scripts/db_times.py (tknnDbscan on BASELINE config 3), average time per kernel
# -> gpurun_out/prof_db/kernels.txt
out=$PWD/gpurun_out/prof_db; rm -rf $out; mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/scripts/db_times.py 10000000 6 > $out/stats.log 2>&1 || { tail -20 $out/stats.log; exit 1; }
cd $root
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_db/stats/**/*kernel_stats.csv", recursive=True)[0]
with open("gpurun_out/prof_db/kernels.txt", "w") as o:
    for r in list(csv.DictReader(open(f)))[:32]:
        line = "%s %s %.1f us  %s %%" % (r["Name"][:72].ljust(72), r["Calls"].rjust(4), float(r["AverageNs"]) / 1000, r["Percentage"])
        print(line); o.write(line + "\n")
PY
