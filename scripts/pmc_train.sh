#!/bin/bash
# rocprofv3 counter passes over the TRAINING step (scripts/bench_train.py, 1024-ray batch): HBM bytes and matrix-pipe occupancy
# of the training forward, the activation-gradient kernel and the weight-gradient GEMMs.  One --pmc set per run, kernel-trace only.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${PMC_OUT:-gpurun_out/pmc_train}
STEPS=${STEPS:-4}
mkdir -p $OUT
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python scripts/bench_train.py --steps $STEPS --warmup 2 > $OUT/$name.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM
run tcc TCC_HIT_sum TCC_MISS_sum
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python scripts/bench_train.py --steps $STEPS --warmup 2 > $OUT/trace.log 2>&1
python scripts/pmc_train_reduce.py $OUT $((STEPS + 2)) > $OUT/summary.json 2> $OUT/reduce.err
find $OUT -name "*.csv" -size +1M -delete
cat $OUT/summary.json
