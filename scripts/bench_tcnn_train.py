"""Hash-grid field (config 5) training step on one GPU: render_rays (64 + 128 samples, train mode) + MSE + backward
+ Adam, 1024-ray batches (opt.py:batch_size) -- ms per step and samples/s.  Usage: python scripts/bench_tcnn_train.py"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mirror_nerf_amd as M  # noqa: E402
from oracle import mirror_nerf_oracle as O  # noqa: E402  (ray generator only)

ap = argparse.ArgumentParser()
ap.add_argument("--rays", type=int, default=1024)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--f16-grads", action="store_true", help="table gradient of the big hashed levels in half2 by packed atomics")
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
models = {k: M.MirrorNeRFTcnn(encoding="hashgrid", bound=6.0, predict_normal=True, predict_mirror_mask=True).to(dev)
          for k in ("coarse", "fine")}
for m_ in models.values():
    m_.table_grad_f16 = a.f16_grads
emb = {"xyz": M.Embedding(0), "dir": M.Embedding(0)}
allrays = torch.from_numpy(O.synthetic_rays(800, 800)).to(dev)
opt = torch.optim.Adam([p for m in models.values() for p in m.parameters()], lr=5e-4, fused=os.environ.get("MNRF_ADAM_FUSED", "1") == "1")
target = torch.rand(a.rays, 3, device=dev)


def step(it):
    idx = torch.randint(0, allrays.shape[0], (a.rays,), device=dev)
    res = M.render_rays(models, emb, allrays[idx], 64, False, 1, 1, 128, compute_normal=False)
    loss = ((res["rgb_coarse"] - target) ** 2).mean() + ((res["rgb_fine"] - target) ** 2).mean() \
        + 0.1 * ((res["mirror_mask_fine"] - 0.5) ** 2).mean() + 1e-4 * res["surface_normal_fine"].pow(2).sum(-1).mean()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    return loss


for it in range(3):
    step(it)
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(a.steps):
    loss = step(it)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
print(f"tcnn train step: {dt * 1e3:.2f} ms  ({a.rays} rays, {a.rays * 256 / dt / 1e6:.1f} M sample evaluations/s fwd+bwd)  loss {float(loss.detach()):.4f}")
