#!/bin/bash
# Developer probe (GPU box): the unchanged reference sample through the OWL runtime, with kernel stats.
set -e
n=${1:-1000000}
python3 - <<PY
import numpy as np, sys
sys.path.insert(0, '.')
from owlraytracing_amd import datasets
np.savetxt('/tmp/pts_$n.csv', datasets.uniform3d($n, seed=0), fmt='%.9g', delimiter=',')
print(datasets.start_radius($n, 10))
PY
r0=$(python3 -c "import sys; sys.path.insert(0,'.'); from owlraytracing_amd import datasets; print(repr(datasets.start_radius($n,10)))")
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sample -- ./oracle/_ref/sample01-trueknn /tmp/pts_$n.csv $n 3 $r0 10 /tmp/time.txt > gpurun_out/sample.log 2>&1 || { tail gpurun_out/sample.log; exit 1; }
grep -iE "time|Round" gpurun_out/sample.log | tail -8
f=$(find gpurun_out/prof_sample -name "*kernel_stats.csv" | head -1)
head -6 "$f" | cut -c1-160
