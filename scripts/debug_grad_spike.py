#!/usr/bin/env python3
"""Debug aid: are the rare large gradients of the fine model during scene training (they precede a collapse of the run) REAL?  Train
as scripts/train_scene.py does (host-driven route, FlatAdam); whenever a batch's fine-model gradient is unusually large, recompute
the same batch's gradients on the static route and with the exact fp32 kernels and print how far they are from each other."""
import os
import sys
import warnings
from types import SimpleNamespace

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
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 3
hp = training.default_hparams(N_importance=64, train_geometry_stage_end_epoch=4, model_type="nerf", bound=4.0, N_emb_xyz=10, N_emb_dir=4)
system = M.NeRFSystem(hp).to(dev)
rays, rgbs, masks = SC.scene_views(48, 100, 100)
rays_t, rgbs_t, masks_t = (torch.from_numpy(x).to(dev) for x in (rays, rgbs, masks))
opt = training.FlatAdam(list(system.models.values()), lr=5e-4)
gamma = 0.1 ** (1.0 / 6000)
loss_fn = training.total_loss_fn(SimpleNamespace(model_type="nerf"), epoch=5)
g = torch.Generator(device=dev).manual_seed(SEED)


def grads_of(r, c, m, static, precision=None, seed=1234):
    """Gradients of this batch (no optimizer step) with fixed draws."""
    torch.manual_seed(seed)
    old = {}
    if precision:
        for mod in system.models.values():
            old[mod] = mod.__dict__.get("_mnrf_precision")
            mod.__dict__["_mnrf_precision"] = precision
    try:
        ex = dict(training.extra_info(system.hparams, m, 5), _guard=False)
        if static:
            ex.update(_static=True, _gt_valid=True)
        system.zero_grad(set_to_none=True)
        res = system(r, ex)
        loss = loss_fn(res, c, m, r) if getattr(loss_fn, "needs_rays", False) else loss_fn(res, c, m)
        loss.backward()
        out = [torch.cat([q.grad.reshape(-1) for q in params_of(mod)]).clone() for mod in system.models.values()]
        words = [int(mod.__dict__["_mnrf_packed"].packed[-1:].view(torch.int32).item()) for mod in system.models.values()]
    finally:
        for mod, v in old.items():
            if v is None:
                mod.__dict__.pop("_mnrf_precision", None)
            else:
                mod.__dict__["_mnrf_precision"] = v
        system.zero_grad(set_to_none=True)
    return float(loss), out, words


with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    hist = []
    for it in range(steps):
        idx = torch.randint(0, rays_t.shape[0], (1024,), device=dev, generator=g)
        r, c, m = rays_t[idx].contiguous(), rgbs_t[idx].contiguous(), masks_t[idx].contiguous()
        l0, g0, w0 = grads_of(r, c, m, False)
        gm = float(g0[1].abs().max())
        med = sorted(hist[-200:])[len(hist[-200:]) // 2] if hist else gm
        hist.append(gm)
        if gm > 8 * med and it > 50:
            l1, g1, w1 = grads_of(r, c, m, True)
            l2, g2, w2 = grads_of(r, c, m, False, precision="fp32")
            rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-30))  # noqa: E731
            print(f"step {it}: fine |g|max {gm:.3g} (median {med:.3g}); loss host {l0:.5f} static {l1:.5f} fp32 {l2:.5f}; guard words {w0} {w1} {w2}; "
                  f"host vs static {rel(g0[1], g1[1]):.2e}, host vs fp32 {rel(g0[1], g2[1]):.2e}, coarse host vs fp32 {rel(g0[0], g2[0]):.2e}; "
                  f"red {[mod.__dict__.get('_mnrf_seed_reduction', 0) for mod in system.models.values()]}", flush=True)
        loss = training.train_step(system, opt, r, c, m, loss_fn, epoch=5)
        opt.param_groups[0]["lr"] *= gamma
        if it % 200 == 0:
            print(it, f"loss {float(loss):.4f} fine gmax {gm:.3g}", flush=True)
