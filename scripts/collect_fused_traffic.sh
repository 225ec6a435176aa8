#!/bin/bash
# HBM-side traffic and kernel time of the fused FIR-stage kernel (option fir_impl=2) at 16,384 x 12, for the record next to
# the two-kernel path:   scripts/collect_fused_traffic.sh   (through gpurun, from the repo root)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="--workload frontend --channels 16384 --blocks 12 --option fir_impl=2 --no-cpu-baseline --no-fir-stage --no-noisy"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_fused -- python3 $R/bench.py $A > $O/bench_trace_fused.log 2>&1
python3 $R/scripts/prof_summary.py $O/trace_fused > $O/kernel_stats_fused_16384x12.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${c}_fused -- python3 $R/bench.py $A --steps 3 --warmup 1 > $O/bench_pmc_${c}_fused.log 2>&1
  python3 $R/scripts/pmc_summary.py $O/pmc_${c}_fused > $O/pmc_${c}_fused_16384x12.txt
done
rm -rf $O/trace_fused $O/pmc_FETCH_SIZE_fused $O/pmc_WRITE_SIZE_fused
echo done
