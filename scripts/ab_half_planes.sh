#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for r in 1 2 3; do
  python scripts/bench_train.py --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('exact ', round(d['ms_per_step'],3), d['route'])"
  MNRF_DW_PLANES_HALF=1 python scripts/bench_train.py --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('y-half', round(d['ms_per_step'],3), d['route'])"
done
export TMPDIR=/tmp
for L in exact half; do
  D=/tmp/prof_$L; rm -rf $D
  if [ $L == half ]; then export MNRF_DW_PLANES_HALF=1; else unset MNRF_DW_PLANES_HALF; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -o p -- python scripts/bench_train.py --steps 40 --warmup 5 > /dev/null 2>&1
  F=$(find $D -name "*kernel_stats.csv" | head -1)
  echo "== kernel stats ($L)"; [ -n "$F" ] && head -5 "$F" | cut -c1-160 && cp "$F" gpurun_out/r06_train_${L}_planes_kernel_stats.csv
done
