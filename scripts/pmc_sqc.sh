#!/bin/bash
# Scalar-cache / instruction-cache / level counters for the kernels of one bench workload (own --pmc runs, no trace domains):
#   scripts/pmc_sqc.sh <tag> [bench args...]      summaries in gpurun_out/pmc_sqc_<tag>.txt
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmcc_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQC_DCACHE_BUSY_CYCLES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_SMEM SQ_BUSY_CU_CYCLES" \
           "SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_BRANCH SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES" \
           "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQC_TC_REQ SQC_TC_STALL SQ_IFETCH_LEVEL"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d $O/pass$i -- python3 $R/bench.py --no-cpu-baseline --no-fir-stage --no-noisy --steps 3 --warmup 1 "$@" > $O/bench_pass$i.log 2>&1 || echo "pass $i failed"
  python3 $R/scripts/pmc_summary.py $O/pass$i > $O/summary_pass$i.txt || true
  rm -rf $O/pass$i
done
cat $O/summary_pass*.txt > $R/gpurun_out/pmc_sqc_$TAG.txt
echo done
