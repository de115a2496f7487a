#!/usr/bin/env python3
"""Fixture G16: --N_emb_xyz 6 --N_emb_dir 2 (opt.py:35-46; 39 / 15 input channels instead of 63 / 27) through the reference's
`train.NeRFSystem.forward` (coarse + fine pass, GT mirror mask, compacted reflected rays, blend): the result dict, the loss and
the reference's autograd gradients of every parameter -- the HIP path evaluates such a model on the kernels built for 10 / 4
bands with the absent bands' weight columns packed as zeros (mirror_nerf_amd/weights.py: canonical).

Build-container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_nemb.py"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

import make_golden as MG  # noqa: E402  (installs the reference import stubs)
import torch  # noqa: E402
from make_golden_loss import first_order_loss, grad_summary  # noqa: E402

W, O, R = MG.W, MG.O, MG.R
N_XYZ, N_DIR = 6, 2


def case(name, n_rays):
    import train as ref_train
    hp = R.get_hparams(predict_normal=True, predict_mirror_mask=True, trace_secondary_rays=True, N_samples=64, N_importance=64,
                       perturb=0, noise_std=0, only_trace_rays_in_mirrors=True, max_recursive_level=1,
                       N_emb_xyz=N_XYZ, N_emb_dir=N_DIR)
    torch.manual_seed(0)
    system = ref_train.NeRFSystem(hp)
    in_xyz, in_dir = 6 * N_XYZ + 3, 6 * N_DIR + 3
    assert system.nerf_coarse.in_channels_xyz == in_xyz and system.nerf_coarse.in_channels_dir == in_dir
    assert system.embedding_xyz.N_freqs == N_XYZ and system.embedding_dir.N_freqs == N_DIR
    sds = W.make_state_dict(0, 2, in_xyz=in_xyz, in_dir=in_dir)
    for mod, sd in zip((system.nerf_coarse, system.nerf_fine), sds):      # the seed rebuild is the reference's construction
        for k, v in mod.state_dict().items():
            assert np.array_equal(v.detach().numpy(), sd[k]), k
        W.apply_tweaks(sd, W.OPAQUE)
        mod.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    system.train_dataset = types.SimpleNamespace(white_back=False)
    rays = MG.pick_rays(n_rays, 21)
    rs = np.random.RandomState(211)
    gt = (rs.uniform(size=n_rays) < 0.3).astype(np.float32)
    target = rs.uniform(size=(n_rays, 3)).astype(np.float32)
    extra = {"mirror_mask": torch.from_numpy(gt.copy()), "is_eval": False, "train_geometry_stage": False}
    res = system(torch.from_numpy(rays), extra)
    loss = first_order_loss(res, torch.from_numpy(target), torch.from_numpy(gt))
    loss.backward()
    ref = MG.to_np(res)
    outs = dict(ref)
    outs["loss"] = np.array(loss.item())
    for mname, mod in (("coarse", system.nerf_coarse), ("fine", system.nerf_fine)):
        for pn_, p_ in mod.named_parameters():
            outs[f"grad__{mname}__{pn_}"] = grad_summary(p_.grad, p_)
    hp_o = dict(N_samples=64, use_disp=False, perturb=0, noise_std=0, N_importance=64, chunk=hp.chunk,
                trace_secondary_rays=True, only_one_field=False, max_recursive_level=1,
                only_trace_rays_in_mirrors=True, for_vis=False)
    orc = O.render_train({"coarse": sds[0], "fine": sds[1]}, {"xyz": N_XYZ, "dir": N_DIR}, rays, hp_o,
                         {"mirror_mask": gt.copy(), "is_eval": False, "train_geometry_stage": False})
    MG.report(name, ref, orc, skip=("z_vals_fine", "weights_fine", "pred_normal_fine", "normal_fine"))
    n_refl = int((ref["rgb_fine_direct"] != ref["rgb_fine"]).any(-1).sum())
    print(f"    rays whose colour changed by reflection: {n_refl}/{n_rays}; loss {loss.item():.6f}")
    assert n_refl > 0
    # the reference's own fp32-vs-fp64 noise on the forward keys and on the gradients
    g32 = {f"{mn}.{pn_}": (p_.grad.clone() if p_.grad is not None else torch.zeros_like(p_))
           for mn, mod in (("coarse", system.nerf_coarse), ("fine", system.nerf_fine)) for pn_, p_ in mod.named_parameters()}
    system.double()
    system.zero_grad()
    res64 = system(torch.from_numpy(rays).double(), {"mirror_mask": torch.from_numpy(gt.copy()).double(), "is_eval": False,
                                                     "train_geometry_stage": False})
    first_order_loss(res64, torch.from_numpy(target).double(), torch.from_numpy(gt).double()).backward()
    floor = {k: float(np.max(np.abs(v.detach().numpy() - ref[k].astype(np.float64)))) if ref[k].size else 0.0 for k, v in res64.items()}
    gfloor = 0.0
    for mn, mod in (("coarse", system.nerf_coarse), ("fine", system.nerf_fine)):
        for pn_, p_ in mod.named_parameters():
            if p_.grad is not None and p_.grad.abs().max() > 0:
                gfloor = max(gfloor, float((g32[f"{mn}.{pn_}"].double() - p_.grad).abs().max() / p_.grad.abs().max()))
    print(f"    reference fp32-vs-fp64: forward floor > 2e-5 {({k: f'{v:.1e}' for k, v in floor.items() if v > 2e-5})}, gradient floor {gfloor:.2e}")
    meta = dict(seed=0, n_models=2, tweaks=W.OPAQUE, checksum=[W.checksum(s) for s in sds], hp=hp_o, grad_floor=gfloor, floor=floor,
                loss="first_order_loss", in_xyz=in_xyz, in_dir=in_dir, N_emb_xyz=N_XYZ, N_emb_dir=N_DIR, is_eval=False, gt_mode="gt30")
    MG.save(name, meta, {"rays": rays, "gt_mask": gt, "target": target}, outs, keep_per_sample=False)


if __name__ == "__main__":
    case("g16_nemb_6_2_train_grads", 64)
