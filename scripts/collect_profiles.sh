#!/bin/bash
# Collect the judged profile set on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats for the headline (full chain, 16,384 channels x 12 blocks) and the FIR-stage
#   (frontend, 1,024 x 50) workloads, then FETCH_SIZE / WRITE_SIZE in separate --pmc passes (never
#   combined with trace domains).  Summaries land in gpurun_out/prof; copy the ones to be judged into
#   profiles/ with the round prefix.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# the third set: the FIR stage at the headline's per-GPU batch (fir_stage_16384 of the bench line); the fourth: the `noisy` leg's workload
for wl in full frontend frontend_16384x12 full_noisy; do
  case $wl in
    frontend_16384x12) args="--workload frontend --channels 16384 --blocks 12" ;;
    full_noisy) args="--workload full --ebn0 8 --noise-cutoff 6250" ;;
    *) args="--workload $wl" ;;
  esac
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$wl -- python3 $R/bench.py $args --no-cpu-baseline --no-fir-stage --no-noisy > $O/bench_trace_$wl.log 2>&1
  python3 $R/scripts/prof_summary.py $O/trace_$wl > $O/kernel_stats_$wl.txt
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${c}_$wl -- python3 $R/bench.py $args --no-cpu-baseline --no-fir-stage --no-noisy --steps 3 --warmup 1 > $O/bench_pmc_${c}_$wl.log 2>&1
    python3 $R/scripts/pmc_summary.py $O/pmc_${c}_$wl > $O/pmc_${c}_$wl.txt
  done
  rm -rf $O/trace_$wl $O/pmc_FETCH_SIZE_$wl $O/pmc_WRITE_SIZE_$wl
  echo "collected $wl"
done
cd $R
python3 bench.py > $O/bench_full.json 2> $O/bench_full.err
python3 bench.py --workload frontend > $O/bench_frontend.json 2> $O/bench_frontend.err
python3 bench.py --workload frontend --channels 16384 --blocks 12 --no-cpu-baseline > $O/bench_frontend_16384x12.json 2> $O/bench_frontend_16384x12.err
echo done
