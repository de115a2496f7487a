#!/usr/bin/env python3
"""Debug aid: does the lowered gradient scale of the range guard (module._mnrf_seed_reduction, mirror_nerf._lower_gradient_scale)
change the gradients?  Trained weights (an npz of scripts/train_scene.py), the same batches and the same random draws on the
host-driven route: split arithmetic with reduction 0 / 4 / 8 against the exact fp32 kernels."""
import os
import sys
import warnings
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import mirror_nerf_amd as M  # noqa: E402
from mirror_nerf_amd import training  # noqa: E402
from mirror_nerf_amd.weights import params_of  # noqa: E402
import make_golden_trained as SC  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
hp = training.default_hparams(N_importance=64, train_geometry_stage_end_epoch=4, model_type="nerf", bound=4.0, N_emb_xyz=10, N_emb_dir=4)
system = M.NeRFSystem(hp).to(dev)
z = np.load(sys.argv[1])
for mname, mod in (("coarse", system.nerf_coarse), ("fine", system.nerf_fine)):
    mod.load_state_dict({k: torch.from_numpy(z[f"{mname}__{k}"]) for k in mod.state_dict()})
rays, rgbs, masks = SC.scene_views(48, 100, 100)
rays_t, rgbs_t, masks_t = (torch.from_numpy(x).to(dev) for x in (rays, rgbs, masks))
loss_fn = training.total_loss_fn(SimpleNamespace(model_type="nerf"), epoch=5)
g = torch.Generator(device=dev).manual_seed(7)
n_batches = int(sys.argv[2]) if len(sys.argv) > 2 else 60


def grads_of(r, c, m, red, precision=None, static=False, seed=99):
    torch.manual_seed(seed)
    for mod in system.models.values():
        mod.__dict__["_mnrf_seed_reduction"] = red
        if precision:
            mod.__dict__["_mnrf_precision"] = precision
        else:
            mod.__dict__.pop("_mnrf_precision", None)
    ex = dict(training.extra_info(system.hparams, m, 5), _guard=False)
    if static:
        ex.update(_static=True, _gt_valid=True)
    system.zero_grad(set_to_none=True)
    res = system(r, ex)
    loss = loss_fn(res, c, m, r)
    loss.backward()
    out = [torch.cat([q.grad.reshape(-1) for q in params_of(mod)]).clone() for mod in system.models.values()]
    words = [int(mod.__dict__["_mnrf_packed"].packed[-1:].view(torch.int32).item()) for mod in system.models.values()]
    return float(loss), out, words


rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-30))  # noqa: E731
worst = {}
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for it in range(n_batches):
        idx = torch.randint(0, rays_t.shape[0], (1024,), device=dev, generator=g)
        r, c, m = rays_t[idx].contiguous(), rgbs_t[idx].contiguous(), masks_t[idx].contiguous()
        l32, g32, _ = grads_of(r, c, m, 0, "fp32")
        row = {}
        for name, red in (("red0", 0), ("red4", 4), ("red8", 8)):
            l, gg, w = grads_of(r, c, m, red)
            row[name] = (rel(gg[0], g32[0]), rel(gg[1], g32[1]), w)
        for k, v in row.items():
            worst[k] = max(worst.get(k, 0.0), v[0], v[1])
        if any(max(v[0], v[1]) > 1e-2 for v in row.values()) or it % 20 == 0:
            print(it, f"fp32 |g|max coarse {float(g32[0].abs().max()):.3g} fine {float(g32[1].abs().max()):.3g};",
                  {k: (f"{v[0]:.1e}", f"{v[1]:.1e}", v[2]) for k, v in row.items()}, flush=True)
print("worst relative difference to the fp32 kernels' gradients:", {k: f"{v:.2e}" for k, v in worst.items()})
