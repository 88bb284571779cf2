#!/bin/bash
# On the GPU box: working tree vs libowl_mi355x_prev.so, interleaved (AB_N, AB_K, AB_REPS).
for i in 1 2 3; do
  for L in libowl_mi355x_prev.so libowl_mi355x.so; do
    echo -n "$L "; OWL_MI355X_LIB=$PWD/owlraytracing_amd/$L timeout -k 10 200 python scripts/quick_bench.py ${AB_N:-10000000} ${AB_K:-10} 3 ${AB_REPS:-7} 2>&1 | grep kernel= | cut -c1-72
  done
done
