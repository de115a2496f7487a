#!/usr/bin/env python3
"""One line per bench.py output file: headline value, roofline fraction, training legs, config 5 frame, cut legs."""
import json
import sys

for path in sys.argv[1:]:
    d = None
    for line in open(path):
        if line.startswith("{"):
            d = json.loads(line)
    if d is None:
        print(path, "no JSON line")
        continue
    t, h = d.get("train_step", {}), d.get("hash_grid_variant", {})
    print(path, round(d["value"]), round(d["roofline"]["frac"], 4), t.get("ms_per_step"), t.get("with_total_loss", {}).get("ms_per_step"),
          t.get("config3_64_plus_192", {}).get("ms_per_step"), h.get("frame_ms"), d.get("incomplete_legs"))
