#!/usr/bin/env python3
"""Fixtures G15: BASELINE config 1 -- coarse-only (N_importance = 0, 64 samples), one reflection bounce -- THROUGH the
recursion drivers of the reference (`select_type = "coarse"`: train.py:147-151, eval.py:132-172), which G3 (render_rays
alone) does not reach.

Build-container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_config1.py
Cases: `NeRFSystem.forward` with a GT mask (compacted reflections, training dict) and with an invalid GT mask whose
predictions straddle 0.5 (predicted + in-place thresholded mask, eval dict), `batched_inference` with one bounce.
Every fixture is checked against the oracle before it is written (make_golden.report)."""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

import make_golden as MG  # noqa: E402  (installs the reference import stubs)
import torch  # noqa: E402

W, O, R = MG.W, MG.O, MG.R
# W.STRADDLE is tuned for the FINE model of seed 0; the coarse model alone needs another offset (43 of 96 level-0 rays
# above 0.5, none closer than 2.2e-4)
STRADDLE_C1 = [["is_mirror_net.2.weight", "mul", 200.0], ["is_mirror_net.2.bias", "add", 2.38],
               ["sigma.bias", "set", 5.0], ["sigma.weight", "mul", 20.0]]


def train_case(name, n_rays, gt_mode, tweaks, **hp_over):
    import train as ref_train
    hp = R.get_hparams(predict_normal=True, predict_mirror_mask=True, trace_secondary_rays=True, N_samples=64,
                       N_importance=0, perturb=0, noise_std=0, chunk=hp_over.pop("chunk", 32768), **hp_over)
    torch.manual_seed(0)
    system = ref_train.NeRFSystem(hp)
    assert not hasattr(system, "nerf_fine") and list(system.models) == ["coarse"]
    _, sds = MG.ref_models(0, 1, tweaks)
    system.nerf_coarse.load_state_dict({k: torch.from_numpy(v) for k, v in sds[0].items()})
    system.train_dataset = types.SimpleNamespace(white_back=False)
    rays = MG.pick_rays(n_rays, 15)
    rs = np.random.RandomState(151)
    gt = (rs.uniform(size=n_rays) < 0.25).astype(np.float32) if gt_mode == "gt25" else -np.ones(n_rays, dtype=np.float32)
    is_eval = hp_over.get("is_eval", False)
    extra = {"mirror_mask": torch.from_numpy(gt.copy()), "is_eval": is_eval, "train_geometry_stage": False}
    ref = MG.to_np(system(torch.from_numpy(rays), extra))
    hp_o = dict(N_samples=64, use_disp=False, perturb=0, noise_std=0, N_importance=0, chunk=hp.chunk,
                trace_secondary_rays=True, only_one_field=False, max_recursive_level=hp.max_recursive_level,
                only_trace_rays_in_mirrors=hp.only_trace_rays_in_mirrors, for_vis=hp.for_vis)
    orc = O.render_train({"coarse": sds[0]}, MG.EMB_O, rays, hp_o,
                         {"mirror_mask": gt.copy(), "is_eval": is_eval, "train_geometry_stage": False})
    MG.report(name, ref, orc)
    assert "rgb_fine" not in ref and "rgb_coarse_direct" in ref or gt_mode != "gt25", sorted(ref)
    n_refl = int((ref["rgb_coarse_direct"] != ref["rgb_coarse"]).any(-1).sum()) if "rgb_coarse_direct" in ref else 0
    print(f"    rays whose colour changed by reflection: {n_refl}/{n_rays}")
    assert n_refl > 0, "the fixture would not exercise the bounce"
    meta = dict(seed=0, n_models=1, tweaks=tweaks, checksum=[W.checksum(s) for s in sds], hp=hp_o, is_eval=is_eval, gt_mode=gt_mode)
    MG.save(name, meta, {"rays": rays, "gt_mask": gt}, ref, keep_per_sample=False)


def eval_case(name, n_rays, tweaks, chunk=32768):
    import eval as ref_eval
    ref_eval.dataset = types.SimpleNamespace(white_back=False)
    hp = R.get_hparams()
    args = types.SimpleNamespace(**vars(hp))
    args.predict_normal = True
    args.predict_mirror_mask = True
    args.only_one_field = False
    args.max_recursive_level = 1
    args.app_control_mirror_roughness = False
    args.app_reflection_substitution = False
    args.app_place_new_mirror = False
    args.app_reflect_newly_placed_objects = False
    mods, sds = MG.ref_models(0, 1, tweaks)
    rays = MG.pick_rays(n_rays, 16)
    ref = MG.to_np(ref_eval.batched_inference({"coarse": mods[0]}, MG.EMB, torch.from_numpy(rays), 64, 0, False, chunk, args=args,
                                              trace_secondary_rays=True, normal_noise_std=0))
    args_o = dict(predict_normal=True, only_one_field=False, only_one_field_fine_epoch=2, max_recursive_level=1,
                  app_control_mirror_roughness=False, trace_ray_times=0, normal_noise_std=0)
    orc = O.render_eval({"coarse": sds[0]}, MG.EMB_O, rays, 64, 0, False, chunk, args_o)
    MG.report(name, ref, orc)
    n_m = int((ref["mirror_mask_coarse"] > 0.5).sum())
    print(f"    mirror rays at level 0: {n_m}/{n_rays}; keys: {sorted(ref)}")
    assert 0 < n_m < n_rays and np.abs(ref["rgb_coarse_reflect"]).max() > 0
    meta = dict(seed=0, n_models=1, tweaks=tweaks, checksum=[W.checksum(s) for s in sds], args=args_o, N_samples=64,
                N_importance=0, chunk=chunk)
    MG.save(name, meta, {"rays": rays}, ref, keep_per_sample=False)


if __name__ == "__main__":
    train_case("g15_c1_train_gt_compact", 96, "gt25", W.OPAQUE, only_trace_rays_in_mirrors=True, max_recursive_level=1)
    train_case("g15_c1_train_pred_straddle_eval", 96, "invalid", STRADDLE_C1, only_trace_rays_in_mirrors=True,
               max_recursive_level=1, is_eval=True)
    eval_case("g15_c1_eval_l1", 96, STRADDLE_C1)
    eval_case("g15_c1_eval_l1_chunk40", 96, STRADDLE_C1, chunk=40)
