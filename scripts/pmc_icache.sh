#!/bin/bash
# instruction-cache and wait breakdown of the field kernel (both tunings)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc2
mkdir -p $OUT
run() { name=$1; var=$2; shift; shift; MNRF_FIELD_VARIANT=$var rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python scripts/prof_chunk.py --reps 1 > $OUT/$name.log 2>&1; }
for v in s1 s2; do
run ic_$v $v SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH
run sq_$v $v SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS
done
