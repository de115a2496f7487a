#!/bin/bash
# Round 6: the training kernels on the dynamic tile queue against the tree before it (exp_libs/r06_base.so, built from the commit
# in front of the change), alternating on one box; then the per-kernel averages of both from rocprofv3 kernel traces.
#   scripts/ab_train_queue.sh [rounds] > gpurun_out/r06_train_queue_ab.txt
R=${1:-3}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
echo "== colour + mask, graph route: base | queue (default) | new binary with the static grid (MNRF_TRAIN_QUEUE=0)"
for r in $(seq $R); do
  MNRF_LIB=exp_libs/r06_base.so python scripts/bench_train.py --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('base  ', round(d['ms_per_step'],3), d['route'])"
  python scripts/bench_train.py --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('queue ', round(d['ms_per_step'],3), d['route'])"
  MNRF_TRAIN_QUEUE=0 python scripts/bench_train.py --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('static', round(d['ms_per_step'],3), d['route'])"
done
echo "== config-3 schedule (64 + 192)"
for r in $(seq $R); do
  MNRF_LIB=exp_libs/r06_base.so python scripts/bench_train.py --steps 60 --warmup 10 --n-importance 128 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('base  ', round(d['ms_per_step'],3))"
  python scripts/bench_train.py --steps 60 --warmup 10 --n-importance 128 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('queue ', round(d['ms_per_step'],3))"
done
echo "== TotalLoss"
for r in $(seq $R); do
  MNRF_LIB=exp_libs/r06_base.so python scripts/bench_train.py --steps 60 --warmup 10 --loss total 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('base  ', round(d['ms_per_step'],3))"
  python scripts/bench_train.py --steps 60 --warmup 10 --loss total 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('queue ', round(d['ms_per_step'],3))"
done
for L in base queue; do
  echo "== kernel stats ($L)"
  D=/tmp/prof_$L; rm -rf $D
  if [ $L == base ]; then export MNRF_LIB=exp_libs/r06_base.so; else unset MNRF_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -o p -- python scripts/bench_train.py --steps 40 --warmup 5 > /dev/null 2>&1
  F=$(find $D -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && head -12 "$F" | cut -c1-200 && cp "$F" gpurun_out/r06_train_${L}_kernel_stats.csv
done
