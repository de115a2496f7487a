#!/usr/bin/env python3
"""Fixtures G9b: parameter gradients of the reference's train-semantics render under each gradient-steering option
(models/rendering.py:223-247, models/mirror_nerf.py:154-183, train.py:284-289), captured from the reference's own
autograd.  Build-container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_flags.py

Each case = fixture G9 `g9_train_grads_full` (same rays, weights, GT mask; loss `full_loss`, or `mask_focus_loss`
for the two mirror-mask options) with ONE option on; the
forward values are identical to G9's, only the gradients move -- the generator asserts that they DO move with respect to
the flag-less run, so a fixture can never pin a no-op."""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.dont_write_bytecode = True

import _ref_import as R  # noqa: E402

R.install()
import torch  # noqa: E402

import weights as W  # noqa: E402
from make_golden import pick_rays, ref_models, save  # noqa: E402
import make_golden_loss as GL  # noqa: E402
from make_golden_loss import grad_summary  # noqa: E402

CASES = {
    # name: (hparams overrides, extra overrides, system.current_epoch, loss)
    "g9b_detach_mask": (dict(), dict(detach_density_for_mask_loss=True), 0, "mask_focus_loss"),
    "g9b_detach_outside_mirror": (dict(only_trace_rays_in_mirrors=False), dict(detach_density_outside_mirror_for_mask_loss=True), 0,
                                  "mask_focus_loss"),
    "g9b_detach_normal": (dict(), dict(detach_density_for_normal_loss=True), 0, "full_loss"),
    "g9b_detach_ref_color": (dict(detach_ref_color_for_blend=True, train_geometry_stage_end_epoch=4), dict(current_epoch=5), 5,
                             "full_loss"),
}


def run(hp_over, extra_over, epoch, loss_name, n_rays=64, want_floor=False):
    loss_fn = getattr(GL, loss_name)
    import train as ref_train
    hp = R.get_hparams(predict_normal=True, predict_mirror_mask=True, trace_secondary_rays=True, N_samples=64, N_importance=64,
                       perturb=0, noise_std=0, only_trace_rays_in_mirrors=True, max_recursive_level=1)
    for k, v in hp_over.items():
        setattr(hp, k, v)
    torch.manual_seed(0)
    system = ref_train.NeRFSystem(hp)
    _, sds = ref_models(0, 2, W.OPAQUE)
    system.nerf_coarse.load_state_dict({k: torch.from_numpy(v) for k, v in sds[0].items()})
    system.nerf_fine.load_state_dict({k: torch.from_numpy(v) for k, v in sds[1].items()})
    system.train_dataset = types.SimpleNamespace(white_back=False)
    system.current_epoch = epoch
    rays = pick_rays(n_rays, 9)
    rs = np.random.RandomState(99)
    gt = (rs.uniform(size=n_rays) < 0.3).astype(np.float32)
    target = rs.uniform(size=(n_rays, 3)).astype(np.float32)
    extra = {"mirror_mask": torch.from_numpy(gt.copy()), "is_eval": False, "train_geometry_stage": False}
    extra.update(extra_over)
    res = system(torch.from_numpy(rays), extra)
    loss = loss_fn(res, torch.from_numpy(target), torch.from_numpy(gt))
    loss.backward()
    mods = (("coarse", system.nerf_coarse), ("fine", system.nerf_fine))
    grads = {f"grad__{mname}__{pn_}": (p_.grad.clone() if p_.grad is not None else torch.zeros_like(p_))
             for mname, mod in mods for pn_, p_ in mod.named_parameters()}
    floor = 0.0
    if want_floor:      # the reference's own noise: the same step in float64
        system.double()
        system.zero_grad()
        extra64 = dict(extra, mirror_mask=torch.from_numpy(gt.copy()).double())
        res64 = system(torch.from_numpy(rays).double(), extra64)
        loss_fn(res64, torch.from_numpy(target).double(), torch.from_numpy(gt).double()).backward()
        for mname, mod in mods:
            for pn_, p_ in mod.named_parameters():
                g64 = p_.grad if p_.grad is not None else torch.zeros_like(p_)
                if g64.abs().max() > 0:
                    floor = max(floor, float((grads[f"grad__{mname}__{pn_}"].double() - g64).abs().max() / g64.abs().max()))
    hp_o = dict(N_samples=64, use_disp=False, perturb=0, noise_std=0, N_importance=64, chunk=hp.chunk, trace_secondary_rays=True,
                only_one_field=False, max_recursive_level=1, only_trace_rays_in_mirrors=hp.only_trace_rays_in_mirrors, for_vis=False,
                detach_ref_color_for_blend=hp.detach_ref_color_for_blend,
                train_geometry_stage_end_epoch=hp.train_geometry_stage_end_epoch)
    return loss.item(), grads, sds, rays, gt, target, hp_o, floor


def main():
    only = sys.argv[1:]
    base = {}
    for name, (hp_over, extra_over, epoch, loss_name) in CASES.items():
        if only and not any(name.startswith(o) for o in only):
            continue
        bkey = (hp_over.get("only_trace_rays_in_mirrors", True), loss_name)
        if bkey not in base:    # the flag-less run of the same configuration
            base[bkey] = run({"only_trace_rays_in_mirrors": bkey[0]}, {}, 0, loss_name)
        loss0, g0 = base[bkey][0], base[bkey][1]
        loss, g, sds, rays, gt, target, hp_o, floor = run(hp_over, extra_over, epoch, loss_name, want_floor=True)
        assert abs(loss - loss0) <= 1e-6 * max(1.0, abs(loss0)), (name, loss, loss0)      # values do not move
        moved = {k: float((g[k] - g0[k]).abs().max() / max(1e-30, float(g0[k].abs().max()))) for k in g}
        top = sorted(moved.items(), key=lambda kv: -kv[1])[:3]
        assert top[0][1] > 1e-2, f"{name}: the option does not change any gradient ({top})"
        print(f"  {name}: loss {loss:.6f}; reference fp32-vs-fp64 gradient floor {floor:.1e}; gradients that move most vs the flag-less run: "
              + ", ".join(f"{k.replace('grad__', '')} {v:.2f}" for k, v in top))
        outs = {"loss": np.array(loss)}
        outs.update({k: grad_summary(v, v) for k, v in g.items()})
        meta = dict(seed=0, n_models=2, tweaks=W.OPAQUE, checksum=[W.checksum(s) for s in sds], hp=hp_o, loss=loss_name,
                    extra=extra_over, epoch=epoch, moved_top=top, grad_floor=floor)
        save(name, meta, {"rays": rays, "gt_mask": gt, "target": target}, outs)


if __name__ == "__main__":
    main()
