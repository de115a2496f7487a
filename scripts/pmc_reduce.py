#!/usr/bin/env python3
"""Reduce the rocprofv3 --pmc passes of scripts/pmc_passes.sh to per-kernel numbers: for every field kernel the LAST dispatch of
each pass (steady state), counters summed over XCDs/instances.  Prints JSON: {kernel: {counter: value, ...}, ...} plus, under
"traffic", the entries of profiles/traffic.json (HBM bytes per launch = 2 x FETCH_SIZE [gfx950: FETCH_SIZE counts 64 B per
128-B request, MI355X_MICROARCH.md "HBM"] + WRITE_SIZE, both reported in KiB)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
try:      # the commit the measured library was built from (__graft_entry__.build writes it; the GPU box has no .git)
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mirror_nerf_amd", "BUILD_COMMIT")) as fh:
        commit = fh.read().strip()
except OSError:
    commit = None
per = defaultdict(lambda: defaultdict(dict))      # kernel -> counter -> dispatch id -> sum
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].replace(", ", ",")
            if k.startswith(("mf::", "tcnn_")):
                k = "mnrf::" + k      # (the hash-grid kernels live in an anonymous namespace of mnrf_tcnn.hip)
            if "field" not in k and "tcnn" not in k and "dw_gemm" not in k:
                continue
            d = per[k][row["Counter_Name"]]
            d[int(row["Dispatch_Id"])] = d.get(int(row["Dispatch_Id"]), 0.0) + float(row["Counter_Value"])
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util      # (mirror_nerf_amd/source_hash.py on its own: importing the package would load torch)
_spec = importlib.util.spec_from_file_location("source_hash", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                           "mirror_nerf_amd", "source_hash.py"))
SH = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(SH)
out, traffic = {}, {}
for k, counters in per.items():
    out[k] = {c: v[max(v)] for c, v in counters.items()}
    out[k]["source_sha1"] = SH.source_sha1(k)      # the sources these counters describe (tests/test_traffic_cpu.py)
    if "FETCH_SIZE" in out[k] and "WRITE_SIZE" in out[k]:
        hbm = (2 * out[k]["FETCH_SIZE"] + out[k]["WRITE_SIZE"]) * 1024
        traffic[k] = {"hbm_bytes_per_launch": hbm, "fetch_kib": out[k]["FETCH_SIZE"], "write_kib": out[k]["WRITE_SIZE"],
                      "commit": commit, "source_sha1": out[k]["source_sha1"]}
    o = out[k]
    if "SQ_WAVE_CYCLES" in o and "SQ_VALU_MFMA_BUSY_CYCLES" in o:
        o["mfma_busy_frac"] = o["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * o["SQ_WAVE_CYCLES"])
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in o:
                o[c + "_frac"] = o[c] / o["SQ_WAVE_CYCLES"]
print(json.dumps({"kernels": out, "traffic": traffic}, indent=1))
