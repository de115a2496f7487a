#!/bin/bash
# rocprofv3 kernel stats of the training step per route (round 5):  bash scripts/prof_train_routes.sh r05a [color_mask|total]
set -u
TAG=${1:-r05}
LOSS=${2:-color_mask}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
for ROUTE in ${ROUTES:-graph static}; do
  MNRF_TRAIN_ROUTE=$ROUTE rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$ROUTE -o t -- \
      python scripts/bench_train.py --steps 40 --warmup 3 --loss $LOSS > $OUT/train_${LOSS}_${ROUTE}.json 2> $OUT/rocprof_$ROUTE.err
  STATS=$(find $OUT/prof_$ROUTE -name "*kernel_stats.csv" | head -1)
  [ -n "$STATS" ] && head -70 "$STATS" > $OUT/train_${LOSS}_${ROUTE}_kernel_stats.csv
  rm -rf $OUT/prof_$ROUTE
done
ls -la $OUT
