#!/bin/bash
# SQ counter passes (own runs, no trace domains) for the kernels of one bench workload; summaries in gpurun_out/.
#   scripts/pmc_sq.sh <tag> [bench args...]
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d $O/pass$i -- python3 $R/bench.py --no-cpu-baseline --no-fir-stage --no-noisy --no-step12 --steps 3 --warmup 1 "$@" > $O/bench_pass$i.log 2>&1
  python3 $R/scripts/pmc_summary.py $O/pass$i > $O/summary_pass$i.txt
  rm -rf $O/pass$i
done
cat $O/summary_pass1.txt $O/summary_pass2.txt > $R/gpurun_out/pmc_sq_$TAG.txt
echo done
