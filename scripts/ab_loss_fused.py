"""A/B on the GPU: training step with the fused colour + mask loss (default) vs the torch-op version."""
import sys, json
import torch
sys.path.insert(0, ".")
from mirror_nerf_amd import training, synthetic as SY, dist as D
rank, world, dev = D.init_from_env()
rays = SY.device_rays(800, 800, dev)
FUSED, TORCH = training.color_mask_loss, training.color_mask_loss_torch
for name, fn in (("fused", FUSED), ("torch", TORCH)) * 4:
    training.color_mask_loss = fn
    r = training.synthetic_train_bench(dev, rays, 100, 10, 1024)
    print(name, round(r["ms_per_step"], 3), "loss", round(r["loss"], 5))
