#!/usr/bin/env python3
"""stdin: bench.py output; prints the training legs' ms per step (scripts/ab_env_train.sh, quick checks on the GPU box)."""
import json
import sys

d = None
for line in sys.stdin:
    if line.startswith("{"):
        d = json.loads(line)
if d is None:
    print("no line")
else:
    t = d.get("train_step", {})
    print("train", t.get("ms_per_step"), t.get("route"), "routes", t.get("routes_ms_per_step"),
          "total", t.get("with_total_loss", {}).get("ms_per_step"), "config3", t.get("config3_64_plus_192", {}).get("ms_per_step"))
