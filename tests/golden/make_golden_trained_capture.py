#!/usr/bin/env python3
"""Fixtures G11: the REFERENCE evaluated on trained weights (tests/golden/g11_trained_weights.npz, written by
make_golden_trained.py / scripts/train_scene.py).  Build-container only:
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_trained_capture.py
Re-uses the capture routines of make_golden.py (same keys, same noise-floor measurements, same oracle cross-check) with
the seeded random-init weights replaced by the trained pair, plain and with the ROUGH tweaks (weights.ROUGH)."""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.dont_write_bytecode = True

import make_golden as MG  # noqa: E402  (installs the reference import shim)
import make_golden_trained as SC  # noqa: E402
import torch  # noqa: E402
import weights as W  # noqa: E402
from oracle import mirror_nerf_oracle as O  # noqa: E402

WEIGHTS = "g11_trained_weights.npz"


def trained_models(seed, n_models, tweaks):
    """Drop-in for make_golden.ref_models: the trained pair (+ tweaks) as reference modules and numpy state dicts."""
    z = np.load(os.path.join(HERE, WEIGHTS))
    sds = [{k[len(m) + 2:]: z[k].copy() for k in z.files if k.startswith(m + "__")} for m in ("coarse", "fine")[:n_models]]
    mods = []
    for sd in sds:
        W.apply_tweaks(sd, tweaks)
        m = MG.MirrorNeRF(in_channels_xyz=63, in_channels_dir=27, predict_normal=True, predict_mirror_mask=True)
        assert list(m.state_dict()) == O.field_param_names()
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        m.eval()
        mods.append(m)
    return mods, sds


def scene_rays(n, seed, H=None, W_=None):
    """Rays of a held-out view of the training scene (instead of make_golden.pick_rays' synthetic camera)."""
    rays, _, _ = SC.scene_views(3, 64, 64, held_out=True)
    idx = np.random.RandomState(seed).choice(rays.shape[0], n, replace=False)
    return rays[np.sort(idx)]


def psnr_case(name, tweaks, res=48):
    """A whole held-out view through eval.batched_inference (one bounce): the reference's image, the analytic ground
    truth and the reference's PSNR -- the HIP render has to land within 0.1 dB (BASELINE north star)."""
    import eval as ref_eval
    ref_eval.dataset = types.SimpleNamespace(white_back=False)
    hp = MG.R.get_hparams()
    args = types.SimpleNamespace(**vars(hp))
    for k, v in dict(predict_normal=True, predict_mirror_mask=True, only_one_field=False, max_recursive_level=1,
                     app_control_mirror_roughness=False, app_reflection_substitution=False, app_place_new_mirror=False,
                     app_reflect_newly_placed_objects=False).items():
        setattr(args, k, v)
    mods, sds = trained_models(0, 2, tweaks)
    rays, gt, gt_mask = SC.scene_views(1, res, res, held_out=True)
    ref = MG.to_np(ref_eval.batched_inference({"coarse": mods[0], "fine": mods[1]}, MG.EMB, torch.from_numpy(rays), 64, 128, False,
                                              32768, args=args, trace_secondary_rays=True))
    psnr = float(-10 * np.log10(np.mean((ref["rgb_fine"].astype(np.float64) - gt) ** 2)))
    import copy
    m64 = {k: copy.deepcopy(v).double() for k, v in (("coarse", mods[0]), ("fine", mods[1]))}
    ref64 = MG.to_np(ref_eval.batched_inference(m64, MG.EMB, torch.from_numpy(rays).double(), 64, 128, False, 32768, args=args,
                                                trace_secondary_rays=True))
    psnr64 = float(-10 * np.log10(np.mean((ref64["rgb_fine"] - gt) ** 2)))
    floor = {k: float(np.max(np.abs(ref64[k] - ref[k].astype(np.float64)))) for k in ref if ref64[k].shape == ref[k].shape}
    floor_frac = {k: MG.off_fraction(ref64[k], ref[k]) for k in ref if ref64[k].shape == ref[k].shape}
    print(f"  {name}: the reference in float64: {psnr64:.3f} dB; fraction of pixels off by > 1e-4 between its fp32 and fp64 runs: "
          f"{floor_frac['rgb_fine']:.4f} (max {floor['rgb_fine']:.1e})")
    macc = float(((ref["mirror_mask_fine"] > 0.5) == (gt_mask > 0.5)).mean())
    print(f"  {name}: reference PSNR vs analytic ground truth {psnr:.3f} dB, mirror-mask accuracy {macc:.3f}, "
          f"max sigma-weighted opacity {ref['opacity_fine'].max():.3f}")
    meta = dict(seed=0, n_models=2, tweaks=tweaks, checksum=[W.checksum(s) for s in sds], weights_file=WEIGHTS, res=res,
                N_samples=64, N_importance=128, psnr_ref=psnr, psnr_ref_fp64=psnr64, mask_accuracy_ref=macc, floor=floor, floor_frac=floor_frac,
                args=dict(predict_normal=True, only_one_field=False, only_one_field_fine_epoch=2, max_recursive_level=1))
    keep = {k: ref[k] for k in ("rgb_fine", "depth_fine", "mirror_mask_fine", "opacity_fine", "surface_normal_fine")}
    MG.save(name, meta, {"rays": rays, "gt_rgb": gt, "gt_mask": gt_mask}, keep, keep_per_sample=False)


def main():
    only = sys.argv[1:]
    MG.ref_models = trained_models
    MG.pick_rays = scene_rays
    orig_save = MG.save

    def save(name, meta, inputs, outputs, keep_per_sample=True):
        meta = dict(meta, weights_file=WEIGHTS)
        orig_save(name, meta, inputs, outputs, keep_per_sample)
    MG.save = save
    z = np.load(os.path.join(HERE, WEIGHTS))
    print("weights:", json.loads(str(z["meta"])))

    def want(tag):
        return not only or any(tag.startswith(o) for o in only)

    for tag, tweaks in (("trained", []), ("rough", W.ROUGH)):
        if want(f"g11_{tag}_render"):
            MG.render_case(f"g11_{tag}_render_test", 128, 4, 128, tweaks=tweaks, test_time=True, compute_normal=False,
                           keep_per_sample=False)
            MG.render_case(f"g11_{tag}_render_train", 48, 5, 128, tweaks=tweaks, compute_normal=True, keep_per_sample=False)
        if want(f"g11_{tag}_eval"):
            MG.eval_case(f"g11_{tag}_eval_l2", 96, tweaks, 2, want_floor=True)
        if want(f"g11_{tag}_psnr"):
            psnr_case(f"g11_{tag}_psnr", tweaks)
    if want("g11_trained_grads"):
        # gradients of a train step (make_golden.grad_case uses weights.OPAQUE by name: the trained pair is opaque already)
        W_OPAQUE, MG.W.OPAQUE = MG.W.OPAQUE, []
        try:
            MG.grad_case("g11_trained_grads_full", 64, MG.full_loss, full_tensors=True)
        finally:
            MG.W.OPAQUE = W_OPAQUE


if __name__ == "__main__":
    main()
