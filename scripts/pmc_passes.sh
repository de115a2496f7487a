#!/bin/bash
# rocprofv3 counter passes for the field kernel, one --pmc set per run (TCC/SQ slot limits,
# MI355X_MICROARCH.md "rocprofv3 PMC slots"); kernel-trace only, no other trace domains.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${PMC_OUT:-gpurun_out/pmc}
mkdir -p $OUT
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python scripts/prof_chunk.py --reps 1 --fused > $OUT/$name.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
find $OUT -name "*.csv" | head -30
