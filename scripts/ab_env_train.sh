#!/bin/bash
# A/B of one environment switch on the training legs of bench.py, alternating runs on one box.
# usage: ab_env_train.sh <out dir under gpurun_out> <ENV_NAME> <value A> <value B> [rounds]
out=gpurun_out/$1; var=$2; a=$3; b=$4; rounds=${5:-2}
mkdir -p "$out"
for i in $(seq 1 "$rounds"); do
  for v in "$a" "$b"; do
    env "$var=$v" MNRF_BENCH_LEGS=train,train_total,config3 python bench.py --steps 1 --warmup 0 > "$out/line_${v}_$i.json" 2> "$out/err_${v}_$i.log"
    python - "$out/line_${v}_$i.json" "$var=$v" <<'P' >> "$out/ab.log"
import json, sys
d = None
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
if d is None:
    print(sys.argv[2], "no line")
else:
    t = d.get("train_step", {})
    print(sys.argv[2], t.get("routes_ms_per_step"), "total", t.get("with_total_loss", {}).get("ms_per_step"),
          "config3", t.get("config3_64_plus_192", {}).get("ms_per_step"))
P
  done
done
cat "$out/ab.log"
