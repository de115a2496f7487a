#!/usr/bin/env python3
"""Fixtures G13: MirrorNeRF WITHOUT the optional heads (models/mirror_nerf.py:80-99: predict_normal=False and/or
predict_mirror_mask=False -- plain NeRF for extract_color_mesh.py:96-108,167-172 and the ablations), captured from the
reference.  Build-container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_variants.py"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.dont_write_bytecode = True

import make_golden as MG  # noqa: E402
import torch  # noqa: E402
import weights as W  # noqa: E402
from oracle import mirror_nerf_oracle as O  # noqa: E402


def models(seed, tweaks, predict_normal, predict_mirror_mask):
    torch.manual_seed(seed)
    mods = [MG.MirrorNeRF(in_channels_xyz=63, in_channels_dir=27, predict_normal=predict_normal,
                          predict_mirror_mask=predict_mirror_mask) for _ in range(2)]
    sds = W.make_state_dict(seed, 2, predict_normal=predict_normal, predict_mirror_mask=predict_mirror_mask)
    for m, sd in zip(mods, sds):
        for k, v in m.state_dict().items():
            assert np.array_equal(v.numpy(), sd[k]), k
        W.apply_tweaks(sd, tweaks)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        m.eval()
    return mods, sds


def render(name, pn, pm, test_time, compute_normal):
    mods, sds = models(0, W.OPAQUE, pn, pm)
    rays = MG.pick_rays(64, 13)
    ctx = torch.enable_grad() if compute_normal else torch.no_grad()
    with ctx:
        ref = MG.to_np(MG.ref_render_rays({"coarse": mods[0], "fine": mods[1]}, MG.EMB, torch.from_numpy(rays), 64, False, 0, 0, 64,
                                          32768, False, test_time, compute_normal=compute_normal))
    orc = O.render_rays({"coarse": sds[0], "fine": sds[1]}, MG.EMB_O, rays, 64, False, 0, 0, 64, 32768, False, test_time,
                        compute_normal=compute_normal)
    MG.report(name, ref, orc, skip=("z_vals_fine", "weights_fine", "pred_normal_fine", "normal_fine", "normal_coarse"))
    meta = dict(seed=0, n_models=2, tweaks=W.OPAQUE, checksum=[W.checksum(s) for s in sds], predict_normal=pn, predict_mirror_mask=pm,
                N_samples=64, N_importance=64, use_disp=False, white_back=False, test_time=test_time, perturb=0, noise_std=0,
                chunk=32768, kwargs=dict(compute_normal=compute_normal), floor={})
    MG.save(name, meta, {"rays": rays}, ref, keep_per_sample=False)


def evaluate(name, pn, pm, tweaks):
    import eval as ref_eval
    ref_eval.dataset = types.SimpleNamespace(white_back=False)
    hp = MG.R.get_hparams()
    args = types.SimpleNamespace(**vars(hp))
    for k, v in dict(predict_normal=pn, predict_mirror_mask=pm, only_one_field=False, max_recursive_level=1,
                     app_control_mirror_roughness=False, app_reflection_substitution=False, app_place_new_mirror=False,
                     app_reflect_newly_placed_objects=False).items():
        setattr(args, k, v)
    mods, sds = models(0, tweaks, pn, pm)
    rays = MG.pick_rays(64, 14)
    ref = MG.to_np(ref_eval.batched_inference({"coarse": mods[0], "fine": mods[1]}, MG.EMB, torch.from_numpy(rays), 64, 64, False, 32768,
                                              args=args, trace_secondary_rays=True))
    import copy
    m64 = {k: copy.deepcopy(v).double() for k, v in (("coarse", mods[0]), ("fine", mods[1]))}
    ref64 = MG.to_np(ref_eval.batched_inference(m64, MG.EMB, torch.from_numpy(rays).double(), 64, 64, False, 32768, args=args,
                                                trace_secondary_rays=True))
    floor = {k: float(np.max(np.abs(ref64[k] - ref[k].astype(np.float64)))) for k in ref if ref64[k].shape == ref[k].shape}
    print("    reference fp32-vs-fp64 floor:", {k: f"{v:.1e}" for k, v in floor.items() if v > 2e-5})
    args_o = dict(predict_normal=pn, only_one_field=False, only_one_field_fine_epoch=2, max_recursive_level=1)
    orc = O.render_eval({"coarse": sds[0], "fine": sds[1]}, MG.EMB_O, rays, 64, 64, False, 32768, args_o)
    MG.report(name, ref, orc, skip=("z_vals_fine", "weights_fine", "pred_normal_fine", "normal_fine"))
    print(f"    mirror rays: {int((ref['mirror_mask_fine'] > 0.5).sum()) if 'mirror_mask_fine' in ref else 'no mask head'}/64")
    meta = dict(seed=0, n_models=2, tweaks=tweaks, checksum=[W.checksum(s) for s in sds], predict_normal=pn, predict_mirror_mask=pm,
                args=args_o, N_samples=64, N_importance=64, chunk=32768, floor=floor)
    MG.save(name, meta, {"rays": rays}, ref, keep_per_sample=False)


def main():
    render("g13_plain_nerf_train", False, False, False, False)
    render("g13_plain_nerf_test", False, False, True, False)
    render("g13_mask_head_only_train", False, True, False, True)
    render("g13_normal_head_only_train", True, False, False, False)
    # no normal head: the reflection uses the density-gradient normal (eval.py:147-148, 338-360)
    evaluate("g13_no_normal_head_eval", False, True, W.STRADDLE)


if __name__ == "__main__":
    main()
