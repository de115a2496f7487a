#!/bin/bash
# A/B of the weight-gradient GEMM variants on one box: correctness (tests/test_hip_backward.py) then step times and the
# per-kernel rocprofv3 view for MNRF_DW_PIPE = 0 (phased kernels), 1 (pipelined 128-wide tiles), 2 (+ 64-wide), hints on/off.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${OUT:-gpurun_out/ab_dw}
CFGS=${CFGS:-"0:1 1:1 1:0 2:1 0:1 1:1"}
mkdir -p $OUT
: > $OUT/summary.txt
timeout 900 python -m pytest tests/test_hip_backward.py -x -q -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/summary.txt
tail -2 $OUT/tests.log | tee -a $OUT/summary.txt
for cfg in $CFGS; do
  pipe=${cfg%%:*}; hint=${cfg##*:}
  for loss in color_mask total; do
    r=$(MNRF_DW_PIPE=$pipe MNRF_DW_HINT=$hint timeout 300 python scripts/bench_train.py --steps 20 --warmup 5 --loss $loss 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))")
    echo "pipe=$pipe hint=$hint loss=$loss ms_per_step=$r" | tee -a $OUT/summary.txt
  done
done
for pipe in ${PROF:-0 1}; do
  MNRF_DW_PIPE=$pipe timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof$pipe -o p -- python scripts/bench_train.py --steps 20 --warmup 3 > $OUT/prof$pipe.log 2>&1
  f=$(find $OUT/prof$pipe -name "*kernel_stats.csv" | head -1)
  echo "== pipe=$pipe" | tee -a $OUT/summary.txt
  head -9 "$f" | cut -c1-150 | tee -a $OUT/summary.txt
  find $OUT/prof$pipe -name "*.csv" -size +1M -delete
done
