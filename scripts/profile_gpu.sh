#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: kernel-trace stats + PMC passes for one command.
#   scripts/profile_gpu.sh <tag> -- python3 scripts/quick_bench.py 10000000 10 2 2
# Output: gpurun_out/prof_<tag>/{stats,pmc_*}/...  (copy the summaries you want judged to profiles/)
set -e
tag=$1; shift; [ "$1" = "--" ] && shift
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
echo "$@" > $out/command.txt
export TMPDIR=/tmp
cd $PWD
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- "$@" > $out/stats.log 2>&1 || { tail -20 $out/stats.log; exit 1; }
# PMC passes (separate runs; never combined with trace domains other than kernel-trace)
i=0
for ctrs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM" \
            "SQ_INSTS_BRANCH SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_LDS_ATOMIC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_THREAD_CYCLES_VALU" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/pmc_$i -- "$@" > $out/pmc_$i.log 2>&1 || { echo "pmc pass $i failed"; tail -5 $out/pmc_$i.log; }
done
python3 scripts/summarize_prof.py $out > $out/summary.txt 2>&1 || true
cat $out/summary.txt
