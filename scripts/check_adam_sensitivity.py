"""How far do three training steps move apart when the initial weights differ by ONE ULP?  (Justifies the tolerance of
test_flat_adam_takes_the_same_steps_as_torch_adam[kernel=True]: mnrf_adam_step agrees with torch's fused Adam to one ulp per step.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mirror_nerf_amd as M  # noqa: E402
from mirror_nerf_amd import synthetic as SY, training  # noqa: E402

DEV = "cuda:0"
rays_all = SY.device_rays(32, 32, DEV)


def run(perturb, kernel=None):
    torch.manual_seed(0)
    system = M.NeRFSystem(training.default_hparams(perturb=0.0, noise_std=0.0)).to(DEV)
    with torch.no_grad():
        for m in system.models.values():
            m.sigma.weight.mul_(20.0)
            m.sigma.bias.fill_(1.0)
        if perturb:
            g = torch.Generator(device=DEV)
            g.manual_seed(99)
            for q in system.parameters():
                q.mul_(1.0 + (torch.randint(0, 3, q.shape, device=DEV, generator=g) - 1).float() * 6e-8)
    opt = torch.optim.Adam(list(system.parameters()), lr=5e-4, fused=True) if kernel is None else \
        training.FlatAdam(list(system.models.values()), lr=5e-4, kernel=kernel)
    g = torch.Generator(device=DEV)
    g.manual_seed(7)
    for _ in range(3):
        idx = torch.randint(0, rays_all.shape[0], (256,), device=DEV, generator=g)
        target = torch.rand(256, 3, device=DEV, generator=g)
        gt = (torch.rand(256, device=DEV, generator=g) < 0.25).float()
        training.train_step(system, opt, rays_all[idx].contiguous(), target, gt)
    torch.cuda.synchronize()
    return {k: v.detach().clone() for k, v in system.named_parameters()}


a, b, c = run(False), run(True), run(False, kernel=True)
for name, other in (("one-ulp perturbation of the initial weights, torch Adam", b), ("mnrf_adam_step, same initial weights", c)):
    worst = max(((a[k] - other[k]).abs().max().item(), (a[k] - other[k]).abs().mean().item(), k) for k in a)
    wm = max(((a[k] - other[k]).abs().mean().item(), k) for k in a)
    print(f"{name}: largest |difference| {worst[0]:.2e} ({worst[2]}), largest mean |difference| {wm[0]:.2e} ({wm[1]})  [lr = 5e-4]")
