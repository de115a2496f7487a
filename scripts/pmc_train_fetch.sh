#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the kernels of a training step (scripts/bench_train.py, 8 steps), for the default library and for
# MNRF_LIB=<variant> -- e.g. the plane stores with the default cache policy (-DMNRF_EXP_STORE_AUX=0) against the non-temporal
# default: the bytes the training forward FETCHES are its weight stream missing the L2.
#   bash scripts/pmc_train_fetch.sh <out dir> [variant.so]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/pmc_train}; LIBV=${2:-}
mkdir -p $OUT
for tag in default variant; do
  if [ $tag == variant ]; then [ -z "$LIBV" ] && continue; export MNRF_LIB=$GRAFT_REPO_ROOT/$LIBV; else unset MNRF_LIB; fi
  for c in ${PMC_COUNTERS:-FETCH_SIZE WRITE_SIZE}; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${tag}_$c -o p -- python scripts/bench_train.py --steps 8 --warmup 2 ${BENCH_ARGS:-} > $OUT/${tag}_$c.log 2>&1
  done
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for tag in ("default", "variant"):
    for c in __import__("os").environ.get("PMC_COUNTERS", "FETCH_SIZE WRITE_SIZE").split():
        files = glob.glob(f"{out}/{tag}_{c}/**/*counter_collection.csv", recursive=True)
        if not files:
            continue
        acc, n = collections.defaultdict(float), collections.Counter()
        for row in csv.DictReader(open(files[0])):
            if row.get("Counter_Name") == c:
                k = row["Kernel_Name"].split("(")[0][-60:]
                acc[k] += float(row["Counter_Value"]); n[k] += 1
        top = sorted(acc.items(), key=lambda kv: -kv[1])[:8]
        print(tag, c, "per launch (KiB for *_SIZE, counts otherwise):", {k: round(v / n[k], 1) for k, v in top}, "launches:", {k: n[k] for k, _ in top})
PY
find $OUT -name "*.csv" -size +1M -delete
