#!/bin/bash
# 48-samples-per-wave tuning (MNRF_SPLIT48=1) of the forward-only split kernels against the default: parity tests, then
# launch times of one 32768-ray chunk alternating between the two on the same box, with the rocm-smi clock under load.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/ab_split48
mkdir -p $OUT
: > $OUT/summary.txt
if [ "${TESTS:-1}" = 1 ]; then
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_hip_guard.py -x -q -m gpu -k "not split32" > $OUT/tests.log 2>&1 < /dev/null
echo "tests rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/tests.log | tee -a $OUT/summary.txt
fi
for v in 0 1 0 1; do
  echo "== MNRF_SPLIT48=$v" | tee -a $OUT/summary.txt
  MNRF_SPLIT48=$v timeout 200 python scripts/prof_chunk.py --reps 6 2>&1 < /dev/null | tail -4 | tee -a $OUT/summary.txt
done
