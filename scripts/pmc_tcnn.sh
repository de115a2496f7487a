#!/bin/bash
# rocprofv3 counter passes for the hash-grid field kernel (one --pmc set per run; kernel-trace only).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PMC_OUT=${PMC_OUT:-gpurun_out/pmc_tcnn}
OUT=$PMC_OUT
mkdir -p $OUT
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python scripts/prof_chunk.py --tcnn --reps 1 > $OUT/$name.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA
run fetch FETCH_SIZE
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob(__import__("os").environ.get("PMC_OUT", "gpurun_out/pmc_tcnn") + "/*/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "tcnn_kernel" not in k: continue
        k = "full" if "Lb0ELb0" in k or "<false, false>" in k else "sigma"
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        n[(k, r["Counter_Name"])] += 1
    for k in acc:
        print(f.split("/")[-2], k, {c: round(v / n[(k, c)]) for c, v in acc[k].items()})
PY
