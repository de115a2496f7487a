#!/bin/bash
# Board power and shader clock (rocm-smi) while the dominant kernels run back to back: split-f16 (default and 32x32x16 tuning)
# and the bit-exact fp32 kernels.  One sample per second over ~8 s of rendering each.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/power_probe
mkdir -p $OUT
rocm-smi --showmaxpower 2>&1 | grep -i "power" | head -3
probe() {
  name=$1; shift
  env "$@" python - > $OUT/$name.run.log 2>&1 < /dev/null <<'PY' &
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench, mirror_nerf_amd as M
from mirror_nerf_amd import mirror_nerf as MN, synthetic as SY
dev = torch.device("cuda", 0)
MN.set_precision(os.environ.get("PREC", "split"))
models, sds, emb = bench.build_models(dev)
rays = SY.device_rays(800, 800, dev)
t0 = time.time()
while time.time() - t0 < 12:
    M.batched_inference(models, emb, rays[:131072], 64, 128, False, 32768, args=bench.ARGS, trace_secondary_rays=True, to_cpu=False)
    torch.cuda.synchronize()
PY
  pid=$!
  sleep 5
  for i in 1 2 3 4 5; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket Graphics Package Power|sclk clock level" | sed "s/^/$name: /"
    sleep 1
  done
  wait $pid
}
probe split PREC=split MNRF_SPLIT32=0
probe split32 PREC=split MNRF_SPLIT32=1
probe fp32 PREC=fp32
