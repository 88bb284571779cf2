#!/bin/bash
# On the GPU box: the packet kernel's phases priced by switching them off in the diagnostic build
# (make -C owlraytracing_amd/csrc DIAG=1).  Results are wrong whenever a bit is set; only times count.
#   scripts/diag_sweep.sh "0 1 2 4 8 64 65" [n] [k]
for d in ${1:-0 1 2 4 8 64}; do
  echo "== TKNN_TEAM_DIAG=$d"
  OWL_MI355X_LIB=$PWD/owlraytracing_amd/libowl_mi355x_diag.so TKNN_TEAM_DIAG=$d timeout -k 10 300 \
    python scripts/quick_bench.py ${2:-10000000} ${3:-10} 3 3 2>&1 | grep "kernel=\|wave-time\|mean busy" | tail -3 | cut -c1-150
done
