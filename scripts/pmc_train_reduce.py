#!/usr/bin/env python3
"""Reduce scripts/pmc_train.sh: per kernel, counters summed over ALL dispatches and XCDs, divided by the number of steps; kernel
time per step from the trace pass.  HBM bytes = 2 x FETCH_SIZE (gfx950 correction, MI355X_MICROARCH.md "HBM") + WRITE_SIZE, KiB."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root, steps = sys.argv[1], int(sys.argv[2])
cnt = defaultdict(lambda: defaultdict(float))
short = lambda n: n.replace("void ", "").split("(")[0].replace(", ", ",")
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = short(row["Kernel_Name"])
            if "mnrf" in k:
                cnt[k][row["Counter_Name"]] += float(row["Counter_Value"])
dur = defaultdict(float)
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            dur[short(row["Kernel_Name"])] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
out = {}
for k, c in cnt.items():
    o = {n: v / steps for n, v in c.items()}
    o["ms_per_step"] = dur.get(k, 0.0) / steps * 1e-6
    if "FETCH_SIZE" in o and "WRITE_SIZE" in o:
        o["hbm_mb_per_step"] = (2 * o["FETCH_SIZE"] + o["WRITE_SIZE"]) * 1024 / 1e6
        if o["ms_per_step"] > 0:
            o["hbm_tb_per_s"] = o["hbm_mb_per_step"] * 1e6 / (o["ms_per_step"] * 1e-3) / 1e12
    if o.get("SQ_WAVE_CYCLES"):
        o["mfma_busy_frac"] = o.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * o["SQ_WAVE_CYCLES"])
        o["wait_any_frac"] = o.get("SQ_WAIT_ANY", 0.0) / o["SQ_WAVE_CYCLES"]
    out[k] = o
print(json.dumps(dict(sorted(out.items(), key=lambda kv: -kv[1]["ms_per_step"])), indent=1))
