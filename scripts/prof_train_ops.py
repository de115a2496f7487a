"""torch.profiler view of one training step: which torch ops launch the small kernels / copies between the HIP kernels."""
import os
import sys
from collections import Counter

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mirror_nerf_amd import synthetic as SY, training  # noqa: E402
from mirror_nerf_amd.recursion import NeRFSystem  # noqa: E402

dev = torch.device("cuda", 0)
rays_all = SY.device_rays(800, 800, dev)
loss_name = sys.argv[1] if len(sys.argv) > 1 else "color_mask"
torch.manual_seed(0)
system = NeRFSystem(training.default_hparams()).to(dev)
with torch.no_grad():
    for m in (system.nerf_coarse, system.nerf_fine):
        m.sigma.weight.mul_(20.0)
        m.sigma.bias.fill_(1.0)
opt = torch.optim.Adam(list(system.parameters()), lr=5e-4, fused=True)
loss_fn = training.total_loss_fn() if loss_name == "total" else training.color_mask_loss
g = torch.Generator(device=dev)
g.manual_seed(1)


def one():
    idx = torch.randint(0, rays_all.shape[0], (1024,), device=dev, generator=g)
    rays = rays_all[idx].contiguous()
    target = torch.rand(1024, 3, device=dev, generator=g)
    gt = (torch.rand(1024, device=dev, generator=g) < 0.25).float()
    return training.train_step(system, opt, rays, target, gt, loss_fn)


for _ in range(3):
    one()
torch.cuda.synchronize()
N = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(N):
        one()
    torch.cuda.synchronize()
ev = prof.events()
# ops that own device kernels / memcpys, with the innermost python frame of this package
cnt, tim = Counter(), Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.kernels:
        frames = [s for s in (e.stack or []) if "mirror_nerf_amd" in s or "scripts/" in s]
        where = frames[0].split("/")[-1] if frames else "?"
        key = (e.name, where)
        cnt[key] += len(e.kernels)
        tim[key] += sum(k.duration for k in e.kernels)
print(f"device launches per step by (op, innermost package frame); {N} steps")
for key, c in cnt.most_common(45):
    print(f"{c / N:7.1f}  {tim[key] / N:8.1f} us  {key[0][:48]:48s} {key[1][:70]}")
