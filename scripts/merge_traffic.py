#!/usr/bin/env python3
"""Merge a pmc_summary.json (scripts/pmc_reduce.py) into profiles/traffic.json: the `traffic` entries per kernel at the top
level (what bench.py's `roofline.traffic` reads), the raw counters per kernel under "pmc" (what `gather_roofline` reads).
usage: merge_traffic.py <pmc_summary.json> <source label> [kernel name filter] [key suffix]
The key suffix (e.g. " [half2 table]") files a pass taken under another configuration of the same kernel as its own entry."""
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(root, "profiles", "traffic.json")
summ = json.load(open(sys.argv[1]))
label = sys.argv[2]
filt = sys.argv[3] if len(sys.argv) > 3 else ""
suffix = sys.argv[4] if len(sys.argv) > 4 else ""
t = json.load(open(path))
t.setdefault("pmc", {})
for k, e in summ["traffic"].items():
    if filt in k:
        t[k + suffix] = dict(e, source=label)
for k, c in summ["kernels"].items():
    if filt in k:
        t["pmc"][k + suffix] = dict(c, source=label)
json.dump(t, open(path, "w"), indent=1)
print("merged", [k for k in summ["kernels"] if filt in k])
