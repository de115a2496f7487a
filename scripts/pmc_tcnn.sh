#!/bin/bash
# rocprofv3 counter passes for the hash-grid field kernels (one --pmc set per run; kernel-trace only), reduced by pmc_reduce.py:
#   bash scripts/pmc_tcnn.sh   ->  $PMC_OUT/pmc_summary.json  (kernels: counters of the last dispatch; traffic: profiles/traffic.json entries)
# HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB; gfx950: FETCH_SIZE counts 64 B per 128-B request, MI355X_MICROARCH.md "HBM").
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PMC_OUT=${PMC_OUT:-gpurun_out/pmc_tcnn}
OUT=$PMC_OUT
mkdir -p $OUT
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python scripts/prof_chunk.py --tcnn --reps 1 > $OUT/$name.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM
python scripts/pmc_reduce.py $OUT > $OUT/pmc_summary.json 2> $OUT/pmc_reduce.err
find $OUT -name "*.csv" -size +2M -delete
python - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ["PMC_OUT"], "pmc_summary.json")))
for k, c in d["kernels"].items():
    line = {q: c[q] for q in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum") if q in c}
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        line["l2_hit_rate"] = c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in c and "TCP_TCC_READ_REQ_sum" in c:
        line["tcp_hit_rate"] = 1.0 - c["TCP_TCC_READ_REQ_sum"] / max(1.0, c["TCP_TOTAL_CACHE_ACCESSES_sum"])
    print(k, line)
PY
