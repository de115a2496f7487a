#!/bin/bash
# Kernel trace (start / end timestamps per dispatch) of the captured training step: shows which launches overlap.
# usage: trace_train_graph.sh <out dir under gpurun_out> [loss]
set -u
TAG=$1; LOSS=${2:-color_mask}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p $OUT
MNRF_TRAIN_ROUTE=graph rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o t -- \
    python scripts/bench_train.py --steps 6 --warmup 2 --loss $LOSS > $OUT/train.json 2> $OUT/rocprof.err
TR=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python - "$TR" > $OUT/trace_tail.txt <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-200:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:10.1f} {(int(r["End_Timestamp"]) - t0) / 1e3:10.1f} q{r.get("Queue_Id", "?"):>3} {r["Kernel_Name"][:90]}')
P
rm -rf $OUT/prof
