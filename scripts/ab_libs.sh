#!/bin/bash
# On the GPU box: several builds of the library interleaved (AB_N, AB_K, AB_REPS, AB_ROUNDS):  scripts/ab_libs.sh prev v72 ""
for i in $(seq 1 ${AB_ROUNDS:-3}); do
  for t in "$@"; do
    L=libowl_mi355x${t:+_$t}.so
    echo -n "$L "; OWL_MI355X_LIB=$PWD/owlraytracing_amd/$L timeout -k 10 200 python scripts/quick_bench.py ${AB_N:-10000000} ${AB_K:-10} 3 ${AB_REPS:-7} 2>&1 | grep kernel= | cut -c1-72
  done
done
