#!/usr/bin/env python3
"""Print the PSNR of the HIP render of the G11 held-out views next to the reference's (fixtures g11_*_psnr), per arithmetic."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mirror_nerf_amd as M  # noqa: E402
from mirror_nerf_amd import metrics, mirror_nerf as MN  # noqa: E402
from tests.golden import fixtures as FX  # noqa: E402

dev = "cuda:0"
for name in ("g11_trained_psnr", "g11_rough_psnr"):
    fx = FX.Fixture(name)
    m = fx.meta
    sds = fx.state_dicts()
    for prec in ("split", "fp32"):
        MN.set_precision(prec)
        models = {}
        for k, sd in zip(("coarse", "fine"), sds):
            mod = M.MirrorNeRF(in_channels_xyz=63, in_channels_dir=27, predict_normal=True, predict_mirror_mask=True)
            mod.load_state_dict({n: torch.from_numpy(v) for n, v in sd.items()})
            models[k] = mod.to(dev)
        emb = {"xyz": M.Embedding(10), "dir": M.Embedding(4)}
        out = M.batched_inference(models, emb, torch.from_numpy(fx.inputs["rays"]).to(dev), m["N_samples"], m["N_importance"], False, 32768,
                                  args=m["args"], trace_secondary_rays=True, to_cpu=False)
        p = float(metrics.psnr(out["rgb_fine"], torch.from_numpy(fx.inputs["gt_rgb"]).to(dev)))
        d = (out["rgb_fine"].cpu() - torch.from_numpy(fx.outputs["rgb_fine"])).abs().max(1)[0]
        print(f"{name} {prec}: HIP {p:.4f} dB, reference {m['psnr_ref']:.4f} dB (fp64 {m['psnr_ref_fp64']:.4f}), delta {p - m['psnr_ref']:+.4f} dB; "
              f"pixels off by > 1e-4: {float((d > 1e-4).float().mean()):.4f} (reference fp32 vs fp64: {m['floor_frac']['rgb_fine']:.4f}), "
              f"guard pinned: {[MN.precision_of(x) for x in models.values()]}")
