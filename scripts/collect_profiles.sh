#!/bin/bash
# Collect the judged profile set on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats for the headline (full chain, 16,384 channels x 16 blocks), the twelve-block step of rounds 1-4,
#   the FIR-stage workloads (1,024 x 50; 16,384 x 16 / 12 / 48) and the 8 dB workload, then FETCH_SIZE / WRITE_SIZE in
#   separate --pmc passes (never combined with trace domains).  Summaries land in gpurun_out/prof; copy the ones to be
#   judged into profiles/ with the round prefix, then scripts/traffic_from_pmc.py gpurun_out/prof <prefix>.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for wl in full full_12 frontend frontend_16384x16 frontend_16384x12 frontend_16384x48 full_noisy; do
  case $wl in
    full_12) args="--workload full --blocks 12" ;;
    frontend_16384x16) args="--workload frontend --channels 16384 --blocks 16" ;;
    frontend_16384x12) args="--workload frontend --channels 16384 --blocks 12" ;;
    frontend_16384x48) args="--workload frontend --channels 16384 --blocks 48 --steps 20" ;;
    full_noisy) args="--workload full --ebn0 8 --noise-cutoff 6250" ;;
    *) args="--workload $wl" ;;
  esac
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$wl -- python3 $R/bench.py $args --no-cpu-baseline --no-fir-stage --no-noisy --no-step12 > $O/bench_trace_$wl.log 2>&1
  python3 $R/scripts/prof_summary.py $O/trace_$wl > $O/kernel_stats_$wl.txt
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${c}_$wl -- python3 $R/bench.py $args --no-cpu-baseline --no-fir-stage --no-noisy --no-step12 --steps 3 --warmup 1 > $O/bench_pmc_${c}_$wl.log 2>&1
    python3 $R/scripts/pmc_summary.py $O/pmc_${c}_$wl > $O/pmc_${c}_$wl.txt
  done
  rm -rf $O/trace_$wl $O/pmc_FETCH_SIZE_$wl $O/pmc_WRITE_SIZE_$wl
  echo "collected $wl"
done
cd $R
python3 bench.py > $O/bench_full.json 2> $O/bench_full.err
echo done
