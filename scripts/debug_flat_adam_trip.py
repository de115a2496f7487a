#!/usr/bin/env python3
"""Debug aid: scene training (as scripts/train_scene.py) with a chosen seed, optimizer and route, printing the loss and each model's
largest gradient entry every 200 steps, the gradient-scale trips of the range guard, and around the first trip whether the flat
gradient buffer still is what the optimizer reads and how far the update moved each model.
usage: debug_flat_adam_trip.py <steps> <batch seed> <kernel|inner|torch> <host|static> [quiet]   (profiles/r05y_train_scene_routes.txt)"""
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
from mirror_nerf_amd import dist as D, training  # noqa: E402
from mirror_nerf_amd.weights import params_of  # noqa: E402
import make_golden_trained as SC  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
KERNEL = (sys.argv[3] != "inner") if len(sys.argv) > 3 else True
ROUTE = sys.argv[4] if len(sys.argv) > 4 else "host"
QUIET = len(sys.argv) > 5
hp = training.default_hparams(N_importance=64, train_geometry_stage_end_epoch=4, model_type="nerf", bound=4.0, N_emb_xyz=10, N_emb_dir=4)
system = M.NeRFSystem(hp).to(dev)
rays, rgbs, masks = SC.scene_views(48, 100, 100)
rays_t, rgbs_t, masks_t = (torch.from_numpy(x).to(dev) for x in (rays, rgbs, masks))
TORCH = len(sys.argv) > 3 and sys.argv[3] == "torch"
opt = torch.optim.Adam(list(system.parameters()), lr=5e-4, fused=True) if TORCH else training.FlatAdam(list(system.models.values()), lr=5e-4, kernel=KERNEL)
gamma = 0.1 ** (1.0 / 6000)
loss_fn = training.total_loss_fn(SimpleNamespace(model_type="nerf"), epoch=5)
g = torch.Generator(device=dev).manual_seed(SEED)
tripped_at = None
with warnings.catch_warnings(record=True) as wlist:
    warnings.simplefilter("always")
    for it in range(steps):
        idx = torch.randint(0, rays_t.shape[0], (1024,), device=dev, generator=g)
        before = [fp.data.clone() for fp in opt.flats] if not TORCH else []
        n_w = len(wlist)
        loss = training.train_step(system, opt, rays_t[idx].contiguous(), rgbs_t[idx].contiguous(), masks_t[idx].contiguous(), loss_fn, epoch=5, gt_valid=True if ROUTE == "static" else None)
        opt.param_groups[0]["lr"] *= gamma
        if len(wlist) > n_w and tripped_at is None:
            tripped_at = it
            print("TRIP reported during step", it, str(wlist[-1].message)[:120])
        if (not QUIET and tripped_at is not None and it <= tripped_at + 6) or it % 200 == 0 or it == steps - 1:
            info = []
            if TORCH:
                info = [dict(gmax=float(max(q.grad.abs().max() for q in params_of(m)))) for m in system.models.values()]
            else:
                for m, fp, b in zip(opt.modules, opt.flats, before):
                    flat = D._flat_bucket(m)
                    gs = [q.grad for q in params_of(m)]
                    alias = flat is not None
                    gcat = torch.cat([x.reshape(-1) for x in gs]) if all(x is not None for x in gs) else None
                    same = bool(alias and gcat is not None and gcat.numel() == flat.numel() and torch.equal(gcat, flat))
                    lay, _ = D._field_layout(m)
                    o, n = lay["is_mirror_net.0.weight"]
                    info.append(dict(alias=alias, same=same, moved=float((fp.data - b).abs().max()), moved_mirror=float((fp.data[o:o + n] - b[o:o + n]).abs().max()),
                                     gmax=float(flat.abs().max()) if flat is not None else None, gmirror=float(flat[o:o + n].abs().max()) if flat is not None else None,
                                     red=m.__dict__.get("_mnrf_seed_reduction", 0), skipped=[int(s.item()) for s in getattr(opt, "_skipped", [])]))
            print(it, f"loss {float(loss):.4f}", info)
