#!/bin/bash
# One-stop evidence run on the GPU box:  bash scripts/profile_round.sh r02d
#   1. python bench.py                                  -> gpurun_out/$TAG/bench.json
#   2. rocprofv3 --kernel-trace --stats of the same bench (no CPU baseline / training legs) -> bench_kernel_stats.csv
#   3. rocprofv3 --pmc passes of one 32768-ray chunk (scripts/pmc_passes.sh: one counter set per run, kernel-trace only)
#   4. scripts/pmc_reduce.py: per-kernel counter sums -> pmc_summary.json (+ the traffic.json entries bench.py reads)
set -u
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
python bench.py --steps ${BENCH_STEPS:-5} --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
STATS=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
[ -n "$STATS" ] && head -40 "$STATS" > $OUT/bench_kernel_stats.csv
PMC_OUT=$OUT/pmc bash scripts/pmc_passes.sh > $OUT/pmc.log 2>&1
python scripts/pmc_reduce.py $OUT/pmc > $OUT/pmc_summary.json 2> $OUT/pmc_reduce.err
rm -rf $OUT/prof
find $OUT/pmc -name "*.csv" -size +2M -delete
ls -la $OUT
