#!/usr/bin/env python3
"""Fixture G18 (round 6): the gradient "spikes" of TotalLoss on trained weights, pinned on the REFERENCE.

DESIGN 6.3 claimed that about one training run in five of scripts/train_scene.py dies because "on trained weights the second-order
terms of TotalLoss throw gradient entries of 20-100 x the median on a few per cent of the batches ... that is the loss's doing
(losses.py:54-78), not the kernels'" -- on the evidence of this package's own fp32 kernels.  Here the reference itself is run:
`train.NeRFSystem.forward` + `losses.TotalLoss` + backward (train.py:102-348, 439-446) on the CPU, in float32 AND float64, on the
committed trained pair (g11_trained_weights.npz) and two batches of the analytic scene that scripts/find_spike_batch.py found on the
GPU (profiles/r06_spike_scan.json): the batch with the LARGEST gradient of the scan and a MEDIAN one.  A batch is named by one
integer (rows np.random.RandomState(i).randint(n_rays, 1024) of make_golden_trained.scene_views(48, 100, 100); perturb =
noise_std = 0), so nothing but integers, losses and gradient digests is stored.

Stored per batch and precision: loss, every loss term, per-tensor gradient norm / abs-max, per-model norm; and the ratios
spike / median.  tests/test_hip_backward.py::test_gradient_spike_is_the_reference_s holds the HIP step to the same ratios.

Build-container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_spike.py [spike_batch median_batch]"""
import json
import os
import sys
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

import make_golden as MG  # noqa: E402  (installs the reference import stubs)
import make_golden_trained as SC  # noqa: E402
import make_golden_trained_capture as C  # noqa: E402
import torch  # noqa: E402

R = MG.R
torch.set_num_threads(8)
EPOCH = 5


def system_for(dtype):
    import train as ref_train
    hp = R.get_hparams(predict_normal=True, predict_mirror_mask=True, trace_secondary_rays=True, N_samples=64, N_importance=64,
                       perturb=0, noise_std=0, only_trace_rays_in_mirrors=True, max_recursive_level=1)
    torch.manual_seed(0)
    system = ref_train.NeRFSystem(hp)
    _, sds = C.trained_models(0, 2, [])
    system.nerf_coarse.load_state_dict({k: torch.from_numpy(v) for k, v in sds[0].items()})
    system.nerf_fine.load_state_dict({k: torch.from_numpy(v) for k, v in sds[1].items()})
    system.train_dataset = types.SimpleNamespace(white_back=False)
    system.train_geometry_stage = False
    if dtype == torch.float64:
        system.double()
        # losses.py:194 casts the GT mask with .float() whatever the prediction's dtype, which nn.BCELoss refuses in a float64 run
        # ("Found dtype Float but expected Double"): the same BCE with the target cast to the prediction's dtype
        class BCE64(torch.nn.Module):
            def forward(self, p, t):
                return torch.nn.functional.binary_cross_entropy(p, t.to(p.dtype), reduction="none")
        system.loss.mirror_mask_loss.loss = BCE64()
    return system


def step(system, batch_id, scene, dtype):
    rays, rgbs, masks = scene
    idx = np.random.RandomState(batch_id).randint(rays.shape[0], size=1024)
    t = lambda a: torch.from_numpy(a[idx].copy()).to(dtype)  # noqa: E731
    b = {"rays": t(rays), "rgbs": t(rgbs), "mirror_mask": t(masks)}
    extra = {"is_eval": False, "mirror_mask": b["mirror_mask"], "only_one_field": False, "only_one_field_fine_epoch": 2,
             "current_epoch": EPOCH, "train_geometry_stage": False,
             "detach_density_outside_mirror_for_mask_loss": False, "detach_density_for_mask_loss": False, "detach_density_for_normal_loss": False}
    system.zero_grad()
    res = system(b["rays"], extra)
    loss, parts = system.loss(res, b, False, EPOCH)
    loss.backward()
    out = {"loss": float(loss), "terms": {k: float(v) for k, v in parts.items()}, "n_mirror": int(masks[idx].sum()), "tensors": {}}
    for mn, mod in (("coarse", system.nerf_coarse), ("fine", system.nerf_fine)):
        sq = 0.0
        for pn, p in mod.named_parameters():
            g = p.grad.double() if p.grad is not None else torch.zeros_like(p).double()
            out["tensors"][f"{mn}.{pn}"] = [float(g.norm()), float(g.abs().max())]
            sq += float((g * g).sum())
        out[f"{mn}_norm"] = sq ** 0.5
    return out


def main():
    if len(sys.argv) > 2:
        spike, median = int(sys.argv[1]), int(sys.argv[2])
    else:
        scan = json.load(open(os.path.join(ROOT, "profiles", "r06_spike_scan.json")))
        spike, median = scan["top"][0]["batch"], scan["median_batch"]["batch"]
    scene = SC.scene_views(48, 100, 100)
    meta = {"spike_batch": spike, "median_batch": median, "epoch": EPOCH, "weights_file": C.WEIGHTS, "views": [48, 100, 100],
            "hp": dict(N_samples=64, N_importance=64, perturb=0, noise_std=0, only_trace_rays_in_mirrors=True, max_recursive_level=1)}
    for name, dtype in (("f32", torch.float32), ("f64", torch.float64)):
        system = system_for(dtype)
        for which, bid in (("spike", spike), ("median", median)):
            t0 = time.time()
            meta[f"{which}_{name}"] = step(system, bid, scene, dtype)
            print(f"  {which} batch {bid} {name}: loss {meta[f'{which}_{name}']['loss']:.6f}  |g| coarse {meta[f'{which}_{name}']['coarse_norm']:.4e} "
                  f"fine {meta[f'{which}_{name}']['fine_norm']:.4e}  [{time.time() - t0:.0f} s]", flush=True)
    for name in ("f32", "f64"):
        for mn in ("coarse", "fine"):
            meta[f"ratio_{mn}_{name}"] = meta[f"spike_{name}"][f"{mn}_norm"] / meta[f"median_{name}"][f"{mn}_norm"]
    print("  spike / median gradient norm:", {k: round(v, 3) for k, v in meta.items() if k.startswith("ratio_")})
    MG.save("g18_grad_spike", meta, {}, {"spike_batch": np.array(spike), "median_batch": np.array(median)})


if __name__ == "__main__":
    main()
