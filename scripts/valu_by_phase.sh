#!/bin/bash
# On the GPU box: VALU / SALU wave-instructions of the packet kernel with phases switched off in the diagnostic build
# (results are wrong whenever a bit is set; only the counts matter):  scripts/valu_by_phase.sh "0 1 2 4 6 8" [n] [k]
export TMPDIR=/tmp
for d in ${1:-0 1 2 4 6 8}; do
  out=$PWD/gpurun_out/valu_phase_$d
  rm -rf $out; mkdir -p $out
  OWL_MI355X_LIB=$PWD/owlraytracing_amd/libowl_mi355x_diag.so TKNN_TEAM_DIAG=$d rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d $out -- python3 scripts/quick_bench.py ${2:-10000000} ${3:-10} 3 2 > $out.log 2>&1
  python3 - $out $d <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "team_kernel" in row["Kernel_Name"]:
            tot[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("diag=%s  " % sys.argv[2] + "  ".join("%s=%.4g" % (c.replace("SQ_INSTS_", ""), sorted(v)[len(v) // 2]) for c, v in sorted(tot.items())))
PY
done
