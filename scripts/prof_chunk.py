#!/usr/bin/env python3
"""One 32768-ray chunk of the bench workload (64 sigma-only coarse + 192 full fine samples per ray),
rendered `--reps` times: the unit that rocprofv3 --pmc passes and tuning experiments are run on.
Prints the HIP-event time of each field launch."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import mirror_nerf_amd as M  # noqa: E402
from mirror_nerf_amd import mirror_nerf as MN  # noqa: E402
from oracle import mirror_nerf_oracle as O  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--rays", type=int, default=32768)
ap.add_argument("--train", action="store_true", help="train-mode forward: full coarse pass + density-gradient normals")
ap.add_argument("--tcnn", action="store_true", help="hash-grid field (config 5) instead of the 8x256 MLP")
ap.add_argument("--fused", action="store_true", help="also render the chunk with the ray-fused fine pass (maps only)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
models, sds, emb = bench.build_models(dev)
if a.tcnn:
    import time
    torch.manual_seed(0)
    models = {k: M.MirrorNeRFTcnn(encoding="hashgrid", bound=6.0, predict_normal=True, predict_mirror_mask=True).to(dev)
              for k in ("coarse", "fine")}
    emb = {"xyz": M.Embedding(0), "dir": M.Embedding(0)}
    rays = torch.from_numpy(O.synthetic_rays(800, 800)[300 * 800: 300 * 800 + a.rays]).to(dev)
    for it in range(a.reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            M.render_rays(models, emb, rays, 64, False, 0, 0, 128, test_time=True, compute_normal=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"tcnn chunk: {dt * 1e3:.2f} ms  {a.rays * 256 / dt / 1e6:.1f} M samples/s  {a.rays / dt / 1e3:.1f} k rays/s")
    sys.exit(0)
rays = torch.from_numpy(O.synthetic_rays(800, 800)[300 * 800: 300 * 800 + a.rays]).to(dev)
MN.LAUNCH_LOG = []
for _ in range(a.reps + 1):
    if a.train:
        with torch.no_grad():   # the train-MODE forward kernels (4 heads + density gradient), without saving activations
            M.render_rays(models, emb, rays, 64, False, 0, 0, 128, test_time=False, compute_normal=True)
    else:
        with torch.no_grad():   # inference kernels (with autograd on, render_rays takes the training forward)
            M.render_rays(models, emb, rays, 64, False, 0, 0, 128, test_time=True, compute_normal=False)
            if a.fused:         # the same chunk, final pass ray-fused (field + compositing in one kernel, no per-sample output)
                M.render_rays(models, emb, rays, 64, False, 0, 0, 128, test_time=True, compute_normal=False, _maps_only=True)
torch.cuda.synchronize()
for flags, B, e0, e1 in MN.LAUNCH_LOG[(4 if a.fused else 2):]:
    ms = e0.elapsed_time(e1)
    flop = B * (MN.FLOP_SIGMA if flags & 1 else MN.FLOP_FULL) + (B * MN.FLOP_GRAD if flags & 2 else 0)
    print(f"flags={flags} B={B} {ms:.3f} ms  {flop / ms / 1e9:.1f} TFLOP/s (algorithmic)")
