#!/bin/bash
# Shader clock under the two tunings of the forward-only split kernel: GRBM_GUI_ACTIVE (busy cycles, summed over the 8 XCDs)
# and the kernel duration of the same dispatch, rocprofv3 --pmc with --kernel-trace only.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/clock_split32
mkdir -p $OUT
for v in 0 1; do
  MNRF_SPLIT32=$v timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/v$v -o p -- python scripts/prof_chunk.py --reps 2 > $OUT/v$v.log 2>&1 < /dev/null
done
python - <<'PY'
import csv, glob, collections
for v in (0, 1):
    cyc = collections.defaultdict(float); dur = {}
    for f in glob.glob(f"gpurun_out/clock_split32/v{v}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "field_split" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                cyc[(r["Dispatch_Id"], r["Kernel_Name"])] += float(r["Counter_Value"])
    for f in glob.glob(f"gpurun_out/clock_split32/v{v}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[(r["Dispatch_Id"], r["Kernel_Name"])] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    for k in sorted(cyc, key=lambda k: int(k[0])):
        if k in dur:
            print(f"MNRF_SPLIT32={v} {k[1].split('(')[0][-44:]:44s} {dur[k] / 1e6:7.3f} ms  GRBM_GUI_ACTIVE/8 = {cyc[k] / 8:.4g}  clock {cyc[k] / 8 / dur[k]:.3f} GHz")
PY
