#!/usr/bin/env python3
"""VERDICT r2 item 8 (exploratory): how many fine samples have a compositing weight of EXACTLY zero -- relu(sigma) = 0 or a
transmittance that has underflowed -- so that their rgb / normal / mirror heads (25 % of a full evaluation's FLOPs) cannot
change any composited output (models/rendering.py:190-213)?  Measured on the trained pair of fixtures G11 (held-out 48x48
view and the 800x800 bench camera) and, for contrast, on the random-init bench weights."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mirror_nerf_amd as M  # noqa: E402
from mirror_nerf_amd import synthetic as SY  # noqa: E402
from tests.golden import fixtures as FX  # noqa: E402

dev = torch.device("cuda", 0)
emb = {"xyz": M.Embedding(10), "dir": M.Embedding(4)}


def module(sd):
    m = M.MirrorNeRF(in_channels_xyz=63, in_channels_dir=27, predict_normal=True, predict_mirror_mask=True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.to(dev)


def measure(models, rays, tag):
    out = {}
    tot = zero = 0
    g16 = g48 = g192 = n16 = n48 = n192 = 0
    for c in range(0, rays.shape[0], 32768):
        with torch.no_grad():
            r = M.render_rays(models, emb, rays[c:c + 32768].contiguous(), 64, False, 0, 0, 128, test_time=True, compute_normal=False)
        w = r["weights_fine"]
        tot += w.numel()
        z = w == 0
        zero += int(z.sum().item())
        # the granularities a kernel could skip at: 16 consecutive samples of a ray (one MFMA column group), 48 (one wave of
        # the 48-samples-per-wave kernel), the whole 192-sample ray (one workgroup of the ray-fused kernel)
        for width in (16, 48, 192):
            a = z.view(z.shape[0], -1, width).all(dim=2)
            if width == 16:
                g16 += int(a.sum().item()); n16 += a.numel()
            elif width == 48:
                g48 += int(a.sum().item()); n48 += a.numel()
            else:
                g192 += int(a.sum().item()); n192 += a.numel()
    out = {"rays": int(rays.shape[0]), "fine_samples": tot, "weight_exactly_zero": zero, "fraction": zero / tot,
           "groups_of_16_all_zero": g16 / n16, "waves_of_48_all_zero": g48 / n48, "rays_of_192_all_zero": g192 / n192}
    print(tag, json.dumps(out))
    return out


fx = FX.Fixture("g11_trained_psnr")
sds = fx.state_dicts()
trained = {"coarse": module(sds[0]), "fine": module(sds[1])}
res = {"trained_heldout_48x48": measure(trained, torch.from_numpy(fx.inputs["rays"]).to(dev), "trained, held-out view"),
       "trained_800x800_bench_camera": measure(trained, SY.device_rays(800, 800, dev)[::7].contiguous(), "trained, bench camera (every 7th ray)")}
rnd, _ = SY.build_models(dev, SY.ALL_MIRROR, seed=0)
res["random_init_bench"] = measure(rnd, SY.device_rays(800, 800, dev)[::7].contiguous(), "random-init bench weights")
res["note"] = ("fraction of fine-pass samples whose compositing weight is exactly 0.0 in fp32 (alpha = 0 because relu(sigma) = 0, or "
               "the transmittance product has underflowed): their heads could be skipped without changing any composited output")
print(json.dumps(res))
