#!/usr/bin/env python3
"""Timeline of ONE training step from a rocprofv3 --kernel-trace CSV: every kernel in issue order with its start offset,
duration and the idle gap in front of it; totals of busy and idle time.  Usage: trace_step.py <kernel_trace.csv> [n_steps]
(the step boundaries are the FusedAdam launches)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
is_adam = ["FusedAdam" in r["Kernel_Name"] or "multi_tensor_apply" in r["Kernel_Name"] for r in rows]
adam = [i for i in range(len(rows) - 1) if is_adam[i] and not is_adam[i + 1]] + ([len(rows) - 1] if is_adam[-1] else [])
if len(adam) < 3:
    sys.exit("need at least three optimizer steps in the trace")
a, b = adam[-2] + 1, adam[-1] + 1          # the last complete step
step = rows[a:b]
t0 = int(rows[a - 1]["End_Timestamp"])
busy = idle = 0
prev_end = t0
out = []
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = max(0, s - prev_end)
    idle += gap
    busy += e - s
    prev_end = max(prev_end, e)
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "")
    name = name.split("(")[0][:60]
    out.append((s - t0, e - s, gap, name))
print(f"step: {len(step)} launches, wall {(prev_end - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us, idle {idle / 1e3:.1f} us")
agg = {}
for off, dur, gap, name in out:
    k = agg.setdefault(name, [0, 0, 0])
    k[0] += 1; k[1] += dur; k[2] += gap
print("per kernel: launches, busy us, idle-before us")
for name, (n, d, g) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print(f"  {n:3d} {d / 1e3:8.1f} {g / 1e3:8.1f}  {name}")
if len(sys.argv) > 2:
    for off, dur, gap, name in out:
        print(f"{off / 1e3:9.1f} {dur / 1e3:8.1f} {gap / 1e3:7.1f}  {name}")
