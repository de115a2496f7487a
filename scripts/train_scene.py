#!/usr/bin/env python3
"""Train the MirrorNeRF pair on the analytic mirror scene of tests/golden/make_golden_trained.py ON THE GPU, through this
package's own training path (NeRFSystem.forward -> losses.TotalLoss -> HIP backward kernels -> Adam), and write the
weights in the reference's state_dict naming.  The weights feed fixtures G11 (parity of the reference and the HIP path
on TRAINED weights): the build container then evaluates the REFERENCE on them (make_golden_trained_capture.py).

    python scripts/train_scene.py --steps 20000 --out gpurun_out/g11_trained_weights.npz [--init tests/golden/g11_trained_weights.npz]
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import mirror_nerf_amd as M  # noqa: E402
from mirror_nerf_amd import training  # noqa: E402
import make_golden_trained as SC  # noqa: E402  (the analytic scene; test infrastructure, not the product)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--lr", type=float, default=5e-4)
    ap.add_argument("--views", type=int, default=48)
    ap.add_argument("--res", type=int, default=100)
    ap.add_argument("--init", default=None, help="npz of a previous run (coarse__*/fine__* arrays) to continue from")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "g11_trained_weights.npz"))
    ap.add_argument("--precision", default="split")
    ap.add_argument("--loss", choices=("total", "color_mask"), default="total")
    ap.add_argument("--epoch", type=int, default=5, help="the epoch number handed to TotalLoss (it gates the loss terms)")
    ap.add_argument("--model", choices=("nerf", "nerf_tcnn"), default="nerf", help="nerf_tcnn: the hash-grid model (BASELINE config 5)")
    ap.add_argument("--bound", type=float, default=4.0, help="nerf_tcnn: half edge of the hash grid's box")
    ap.add_argument("--table-grad", choices=("fixed", "fp32", "f16"), default="fixed", help="nerf_tcnn: how the table gradient is accumulated")
    ap.add_argument("--flat-adam", action="store_true", help="nerf: training.FlatAdam (mnrf_adam_step) instead of torch's fused Adam")
    ap.add_argument("--mlp-f16", action="store_true", help="nerf_tcnn: single-pass f16 MLPs in the forward kernel")
    ap.add_argument("--route", choices=("host", "static", "graph"), default="host",
                    help="nerf: host = the reference's control flow (reads the reflected-ray count on the host); static = the count stays on the "
                         "device; graph = the whole step replayed as one hipGraph (training.GraphedTrainStep; implies --flat-adam)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    M.set_precision(a.precision)
    torch.manual_seed(0)
    hp = training.default_hparams(N_importance=64, train_geometry_stage_end_epoch=4, model_type=a.model, bound=a.bound,
                                  N_emb_xyz=0 if a.model == "nerf_tcnn" else 10, N_emb_dir=0 if a.model == "nerf_tcnn" else 4)
    system = M.NeRFSystem(hp).to(dev)
    if a.model == "nerf_tcnn":
        for m_ in system.models.values():
            m_.table_grad_fixed, m_.table_grad_f16, m_.mlp_f16 = a.table_grad == "fixed", a.table_grad == "f16", a.mlp_f16
    if a.init:
        z = np.load(a.init)
        for mname, mod in (("coarse", system.nerf_coarse), ("fine", system.nerf_fine)):
            mod.load_state_dict({k: torch.from_numpy(z[f"{mname}__{k}"]) for k in mod.state_dict()})
    rays, rgbs, masks = SC.scene_views(a.views, a.res, a.res)
    rays_t, rgbs_t, masks_t = (torch.from_numpy(x).to(dev) for x in (rays, rgbs, masks))
    vr, vc, vm = SC.scene_views(1, 64, 64, held_out=True)
    vr_t, vc_t = torch.from_numpy(vr).to(dev), torch.from_numpy(vc).to(dev)
    gamma = 0.1 ** (1.0 / max(1, a.steps))
    if a.route == "graph":
        a.flat_adam = True
    if a.flat_adam and a.model == "nerf":
        opt = training.FlatAdam(list(system.models.values()), lr=a.lr)

        class sched:        # (FlatAdam reads its learning rate from param_groups at every step: the same exponential decay)
            @staticmethod
            def step():
                opt.param_groups[0]["lr"] *= gamma
    else:
        opt = torch.optim.Adam(list(system.parameters()), lr=a.lr, fused=True)
        sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=gamma)
    loss_fn = training.total_loss_fn(SimpleNamespace(model_type=a.model), epoch=a.epoch) if a.loss == "total" else training.color_mask_loss
    g = torch.Generator(device=dev).manual_seed(1)
    emb = system.embeddings
    args = dict(predict_normal=True, only_one_field=False, only_one_field_fine_epoch=2, max_recursive_level=1)

    def val():
        out = M.batched_inference(system.models, emb, vr_t, 64, 64, False, 32768, args=args, trace_secondary_rays=True, to_cpu=False)
        mse = float(((out["rgb_fine"] - vc_t) ** 2).mean())
        macc = float(((out["mirror_mask_fine"] > 0.5).float().cpu().numpy() == vm).mean())
        return -10 * np.log10(mse), macc

    graphed = training.GraphedTrainStep(system, opt, a.batch, loss_fn, epoch=a.epoch, gt_valid=True) if (a.route == "graph" and a.model == "nerf") else None
    t0 = time.time()
    for it in range(a.steps):
        idx = torch.randint(0, rays_t.shape[0], (a.batch,), device=dev, generator=g)
        if graphed is not None:
            loss = graphed(rays_t[idx], rgbs_t[idx], masks_t[idx])
        else:
            loss = training.train_step(system, opt, rays_t[idx].contiguous(), rgbs_t[idx].contiguous(), masks_t[idx].contiguous(), loss_fn, epoch=a.epoch,
                                       gt_valid=True if a.route == "static" else None)      # (the analytic scene's masks are 0 / 1: valid)
        sched.step()
        if it % 1000 == 0 or it == a.steps - 1:
            p, macc = val()
            with torch.no_grad():
                tr = M.batched_inference(system.models, emb, rays_t[idx].contiguous(), 64, 64, False, 32768, args=args, trace_secondary_rays=True, to_cpu=False)
                ptr = -10 * np.log10(float(((tr["rgb_fine"] - rgbs_t[idx]) ** 2).mean()))
            print(f"step {it:6d}  loss {loss.item():.4f}  train-batch psnr {ptr:.2f}  held-out psnr {p:.2f} dB  mirror-mask accuracy {macc:.3f}  [{time.time()-t0:.0f} s]", flush=True)
    arrs = {}
    for mname, mod in (("coarse", system.nerf_coarse), ("fine", system.nerf_fine)):
        for k, v in mod.state_dict().items():
            if v.numel() <= 4_000_000:        # (the hash-grid tables stay on the box)
                arrs[f"{mname}__{k}"] = v.detach().cpu().numpy().copy()
    p, macc = val()
    arrs["meta"] = np.array(json.dumps(dict(steps=a.steps, batch=a.batch, lr=a.lr, views=a.views, res=a.res, init=a.init,
                                            trained_with="mirror_nerf_amd (scripts/train_scene.py) on MI355X, precision " + a.precision,
                                            held_out_psnr=p, mirror_mask_accuracy=macc,
                                            sigma_max=float(max((m.sigma.weight if hasattr(m, "sigma") else m.sigma_net[1].weight[0]).abs().max() for m in system.models.values())))))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    np.savez_compressed(a.out, **arrs)
    print("wrote", a.out, f"{os.path.getsize(a.out)/1e6:.1f} MB; held-out PSNR {p:.2f} dB; ms/step {(time.time()-t0)/a.steps*1e3:.2f}")


if __name__ == "__main__":
    main()
