#!/bin/bash
# Where the energy of the split-f16 field kernel goes: launch time and rocm-smi power / shader clock with one activity
# compiled out (experiment libraries built with -DMNRF_EXP_NO_DMA / NO_READ / NO_CONV on mnrf_field_split.hip; results of
# those are garbage by construction), or with an environment switch of the shipped library (argument NAME=VALUE).  One
# 32768-ray chunk rendered back to back for ~9 s per variant.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/energy
mkdir -p $OUT
for tag in base "$@"; do
  lib=$GRAFT_REPO_ROOT/mirror_nerf_amd/libmnrf_hip.so
  extra="MNRF_NOOP=1"
  case $tag in
    base) ;;
    *=*) extra=$tag ;;                                   # an environment switch of the shipped library, e.g. MNRF_SPLIT48=1
    *) lib=$GRAFT_REPO_ROOT/build_exp/libmnrf_$tag.so ;;
  esac
  env $extra MNRF_LIB=$lib timeout 120 python - > $OUT/$tag.log 2>&1 < /dev/null <<'PY' &
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench, mirror_nerf_amd as M
from mirror_nerf_amd import mirror_nerf as MN, synthetic as SY
dev = torch.device("cuda", 0)
models, sds, emb = bench.build_models(dev)
rays = SY.device_rays(800, 800, dev)[300 * 800:300 * 800 + 32768].contiguous()
MN.LAUNCH_LOG = []
t0 = time.time()
with torch.no_grad():
    while time.time() - t0 < 9:
        M.render_rays(models, emb, rays, 64, False, 0, 0, 128, test_time=True, compute_normal=False)
        torch.cuda.synchronize()
log = MN.LAUNCH_LOG[20:]
full = [e0.elapsed_time(e1) for (f, B, e0, e1) in log if not (f & 1)]
sig = [e0.elapsed_time(e1) for (f, B, e0, e1) in log if (f & 1)]
print(f"full {sum(full) / len(full):.3f} ms  sigma-only {sum(sig) / len(sig):.3f} ms  launches {len(full)}")
PY
  pid=$!
  sleep 5
  p=""; c=""
  for i in 1 2 3; do
    s=$(rocm-smi --showpower --showclocks 2>/dev/null)
    p="$p $(echo "$s" | grep -oE "Package Power \(W\): [0-9.]+" | grep -oE "[0-9.]+$")"
    c="$c $(echo "$s" | grep -oE "sclk clock level: [0-9]+: \([0-9]+Mhz" | grep -oE "[0-9]+Mhz")"
    sleep 1
  done
  wait $pid
  echo "$tag: $(tail -1 $OUT/$tag.log) | power W:$p | sclk:$c"
done
