"""Build container only: how the REFERENCE's own train-step gradients (hash-grid models, fixture G17 set-up) move under tiny
perturbations of the rays."""
import sys, os, types
sys.path.insert(0, "/root/repo/tests/golden"); sys.path.insert(0, "/root/repo")
sys.dont_write_bytecode = True
import numpy as np
import make_golden_tcnn as G
import torch
R, MG = G.R, G.MG
from make_golden_loss import first_order_loss
import train as ref_train

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 41
hp = R.get_hparams(model_type="nerf_tcnn", bound=G.RENDER["bound"], predict_normal=True, predict_mirror_mask=True,
                   trace_secondary_rays=True, N_samples=64, N_importance=64, perturb=0, noise_std=0,
                   only_trace_rays_in_mirrors=True, max_recursive_level=1)
system = ref_train.NeRFSystem(hp)
mods, ws, cfg = G.pair(seed)
for mod, src in ((system.nerf_coarse, mods[0]), (system.nerf_fine, mods[1])):
    mod.load_state_dict(src.state_dict())
system.train_dataset = types.SimpleNamespace(white_back=False)
rays = MG.pick_rays(64, seed)
rs = np.random.RandomState(seed + 2)
gt = (rs.uniform(size=64) < 0.3).astype(np.float32)
target = rs.uniform(size=(64, 3)).astype(np.float32)
t = lambda a: torch.from_numpy(a.copy())


def grads(rays):
    system.zero_grad()
    res = system(t(rays), {"mirror_mask": t(gt), "is_eval": False, "train_geometry_stage": False})
    loss = first_order_loss(res, t(target), t(gt))
    loss.backward()
    return {f"{mn}.{pn}": p.grad.numpy().copy() for mn, mod in (("coarse", system.nerf_coarse), ("fine", system.nerf_fine)) for pn, p in mod.named_parameters() if p.grad is not None}, res


a, ra = grads(rays)
for eps in (1e-7, 1e-6, 1e-5):
    rp = np.random.RandomState(7)
    r2 = rays.copy()
    r2[:, :3] += rp.normal(size=(64, 3)).astype(np.float32) * eps
    b, rb = grads(r2)
    worst = sorted(((np.abs(a[k] - b[k]).max() / np.abs(a[k]).max(), k) for k in a), reverse=True)[:3]
    print(f"origin perturbation {eps:g}: ", [(f"{v:.2e}", k) for v, k in worst],
          " rgb_fine moves", float((ra["rgb_fine"] - rb["rgb_fine"]).abs().max()))
