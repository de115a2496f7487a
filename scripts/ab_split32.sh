#!/bin/bash
# 32x32x16 tuning of the forward-only split kernels against the 16x16x32 one: parity tests with MNRF_SPLIT32=1, then the
# launch times of one 32768-ray chunk (scripts/prof_chunk.py) alternating between the two on the same box.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/ab_split32
mkdir -p $OUT
: > $OUT/summary.txt
MNRF_SPLIT32=1 timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_hip_guard.py -x -q -m gpu > $OUT/tests.log 2>&1 < /dev/null
echo "tests rc=$?" | tee -a $OUT/summary.txt
tail -4 $OUT/tests.log | tee -a $OUT/summary.txt
for v in 0 1 0 1 0 1; do
  echo "== MNRF_SPLIT32=$v" | tee -a $OUT/summary.txt
  MNRF_SPLIT32=$v timeout 200 python scripts/prof_chunk.py --reps 4 2>&1 < /dev/null | tail -4 | tee -a $OUT/summary.txt
done
