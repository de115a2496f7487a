#!/usr/bin/env python3
"""Training-step throughput (BASELINE config 3 shape), one JSON line on rank 0.
Single GPU: python scripts/bench_train.py;  N GPUs: python -m torch.distributed.run --nproc-per-node N ..."""
import argparse
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mirror_nerf_amd import training  # noqa: E402
from mirror_nerf_amd import dist as D, synthetic as SY  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--n-importance", type=int, default=None, help="--N_importance (default 64: run.sh:266; 128 = BASELINE config 3 as worded)")
ap.add_argument("--loss", choices=("color_mask", "total", "run_sh", "run_sh_stage"), default="color_mask",
                help="total = the reference's TotalLoss through the fused HIP loss kernels (reads normal_*: second-order pass on)")
a = ap.parse_args()
rank, world, dev = D.init_from_env()
rays = SY.device_rays(800, 800, dev)
kw = {} if a.n_importance is None else {"N_importance": a.n_importance}
r = training.synthetic_train_bench(dev, rays, a.steps, a.warmup, a.batch, loss_name=a.loss, **kw)
if rank == 0:
    r["n_gpus"] = world
    print(json.dumps(r))
if dist.is_initialized():
    dist.destroy_process_group()
