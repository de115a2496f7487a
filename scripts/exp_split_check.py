#!/usr/bin/env python3
"""fp32 vs split-f16 field kernel on the same samples: max-abs differences per output and the
HIP-event time of a 6.29 M-sample launch of each (experiment behind DESIGN.md section 4.1b)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mirror_nerf_amd import mirror_nerf as MN  # noqa: E402
from oracle import mirror_nerf_oracle as O  # noqa: E402

dev = torch.device("cuda", 0)
models, sds, emb = bench.build_models(dev)
m = models["fine"]
rays = torch.from_numpy(O.synthetic_rays(800, 800)[300 * 800: 300 * 800 + 32768]).to(dev)
z = torch.linspace(0.05, 8.0, 192, device=dev).repeat(32768, 1).contiguous()
de = emb["dir"](rays[:, 3:6].contiguous())
B = 32768 * 192
for sigma_only, grad in ((False, False), (True, False), (False, True)):
    res = {}
    tile_samples = 128
    for mode in ("fp32", "split") + (tuple(os.environ["MNRF_EXP_X"].split(",")) if os.environ.get("MNRF_EXP_X") else ()):
        MN.set_precision(mode)
        MN.LAUNCH_LOG = []
        for _ in range(3):
            o = MN.field_forward(m, B, rays=rays, z_vals=z, spr=192, dir_emb=de, dir_stride=27,
                                 sigma_only=sigma_only, grad_normal=grad, want_geo=bool(os.environ.get("MNRF_EXP_CYCLES")))
        torch.cuda.synchronize()
        ms = min(e0.elapsed_time(e1) for _, _, e0, e1 in MN.LAUNCH_LOG[1:])
        flop = B * ((MN.FLOP_SIGMA if sigma_only else MN.FLOP_FULL) + (MN.FLOP_GRAD if grad else 0))
        print(f"sigma_only={sigma_only} grad={grad} {mode}: {ms:.3f} ms  {flop / ms / 1e9:.1f} TFLOP/s algorithmic  "
              f"{B / ms / 1e3:.1f} M samples/s")
        res[mode] = o
        if os.environ.get("MNRF_EXP_CYCLES") and mode != "fp32":   # library built with -DMNRF_EXP_CYCLES: s_memtime per tile
            mk = o.pop("geo_feat").view(-1).view(torch.int64)[: (B // tile_samples) * 16].view(-1, 16).double()
            names = {1: "prologue", 2: "L1", 3: "L2-4", 4: "L5", 5: "L6-8", 6: "sigma", 7: "normal", 8: "mirror", 9: "final",
                     10: "dir+rgb", 15: "rest (grad pass)"}
            prev, line = 0, []
            for k in sorted(names):
                if sigma_only and 7 <= k <= 10:
                    continue
                line.append(f"{names[k]} {(mk[:, k] - mk[:, prev]).mean().item():.0f}")
                prev = k
            tot = (mk[:, 15] - mk[:, 0]).mean().item()
            print("   cycles per 128-sample tile: " + " | ".join(line) + f" | total {tot:.0f}"
                  f"  -> clock {tot * (B // tile_samples) / 256 / (ms * 1e-3) / 1e9:.3f} GHz if the 256 CUs were always busy")
    for mode in list(res)[1:]:
      for k in res["fp32"]:
        a, b = res["fp32"][k], res[mode][k]
        d = (a - b).abs()
        print(f"   {mode} {k}: max|fp32-split| = {d.max().item():.3e}  mean {d.mean().item():.3e}  (max|fp32| {a.abs().max().item():.3e})"
              f"  finite={bool(torch.isfinite(b).all())}")
