#!/bin/bash
# GPU box: rocprofv3 kernel trace of the reference's unchanged sample (oracle/_ref/sample01-trueknn, 1 M points, k = 10):
# time of the __raygen__ launches with the launch indices in thread order (OWL_LAUNCH_ORDER=0) and in the tree's Morton order
root=$PWD
python3 - <<PY
import numpy as np, sys
sys.path.insert(0, "$root")
from owlraytracing_amd import datasets
n = 1_000_000
np.savetxt("/tmp/pts_%d.csv" % n, datasets.uniform3d(n, seed=0), fmt="%.9g", delimiter=",")
print(repr(datasets.start_radius(n, 10)))
PY
r0=$(python3 -c "import sys; sys.path.insert(0,'$root'); from owlraytracing_amd import datasets; print(repr(datasets.start_radius(1000000,10)))")
cd /tmp && export TMPDIR=/tmp
for o in 0 1; do
  rm -rf /tmp/prof_sample_$o
  OWL_LAUNCH_ORDER=$o rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sample_$o -- $root/oracle/_ref/sample01-trueknn /tmp/pts_1000000.csv 1000000 3 $r0 10 /tmp/time.txt > /tmp/prof_sample_$o.log 2>&1
  echo "OWL_LAUNCH_ORDER=$o"
  python3 - <<PY
import csv, glob
f = glob.glob("/tmp/prof_sample_$o/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print("  %s %s x %.1f us" % (r["Name"][:60].ljust(60), r["Calls"], float(r["AverageNs"]) / 1000))
PY
done
