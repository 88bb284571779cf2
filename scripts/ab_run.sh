#!/bin/bash
# On the GPU box: team-kernel parity tests, then working tree vs prev library at C2, interleaved.
set -e
timeout -k 10 400 python -m pytest tests/test_trueknn_gpu.py -m gpu -x -q -k "team" > gpurun_out/t_team.log 2>&1 || { tail -20 gpurun_out/t_team.log; exit 1; }
tail -1 gpurun_out/t_team.log
for i in 1 2; do
  for L in libowl_mi355x_prev.so libowl_mi355x.so; do
    echo -n "$L "; OWL_MI355X_LIB=$PWD/owlraytracing_amd/$L timeout -k 10 200 python scripts/quick_bench.py ${AB_N:-10000000} ${AB_K:-10} 3 7 2>&1 | grep kernel= | cut -c1-90
  done
done
