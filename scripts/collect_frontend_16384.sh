#!/bin/bash
# kernel-trace stats of the FIR-stage workload at the headline size (front end + timing stage, 16,384 channels x 12 blocks),
# to go with bench_frontend_16384x12.json:   scripts/collect_frontend_16384.sh   (through gpurun, from the repo root)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_fe16k -- python3 $R/bench.py --workload frontend --channels 16384 --blocks 12 --no-cpu-baseline --no-fir-stage --no-noisy > $O/bench_trace_fe16k.log 2>&1
python3 $R/scripts/prof_summary.py $O/trace_fe16k > $O/kernel_stats_frontend_16384x12.txt
rm -rf $O/trace_fe16k
echo done
