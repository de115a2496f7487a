#!/usr/bin/env python3
"""Round 6: find, on the COMMITTED trained pair (tests/golden/g11_trained_weights.npz), training batches of the
analytic mirror scene whose TotalLoss gradient is far larger than a typical batch's -- the "spikes" DESIGN 6.3 attributes to the
loss (losses.py:54-78: NormalLoss on normal_dif_*, x 100 inside the mirror) and not to the kernels.  Deterministic draws
(perturb = noise_std = 0), batch i = rows np.random.RandomState(i).randint(n_rays, size=1024) of make_golden_trained.scene_views(48,
100, 100): a batch is named by ONE integer, so the fixture generator (tests/golden/make_golden_spike.py) can rebuild it on the CPU
and run the REFERENCE on it.

    python scripts/find_spike_batch.py [n_batches] > gpurun_out/r06_spike_scan.json      (GPU box)
"""
import json
import os
import sys
import warnings
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import mirror_nerf_amd as M  # noqa: E402
from mirror_nerf_amd import training  # noqa: E402
from mirror_nerf_amd.weights import params_of  # noqa: E402
import make_golden_trained as SC  # noqa: E402


def main():
    n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    dev = torch.device("cuda", 0)
    hp = training.default_hparams(N_importance=64, perturb=0.0, noise_std=0.0)
    system = M.NeRFSystem(hp).to(dev)
    z = np.load(os.path.join(ROOT, "tests", "golden", "g11_trained_weights.npz"))
    for name, mod in (("coarse", system.nerf_coarse), ("fine", system.nerf_fine)):
        mod.load_state_dict({k[len(name) + 2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(name + "__")})
    system.to(dev)
    rays, rgbs, masks = SC.scene_views(48, 100, 100)
    rays_t, rgbs_t, masks_t = (torch.from_numpy(x).to(dev) for x in (rays, rgbs, masks))
    loss_fn = training.total_loss_fn(SimpleNamespace(model_type="nerf"), epoch=5)
    rows = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i in range(n_batches):
            idx = torch.from_numpy(np.random.RandomState(i).randint(rays.shape[0], size=1024)).to(dev)
            r, c, m = rays_t[idx].contiguous(), rgbs_t[idx].contiguous(), masks_t[idx].contiguous()
            ex = dict(training.extra_info(system.hparams, m, 5), _guard=False)
            system.zero_grad(set_to_none=True)
            res = system(r, ex)
            loss = loss_fn(res, c, m, r)
            loss.backward()
            g = [torch.cat([q.grad.reshape(-1) for q in params_of(mod)]) for mod in system.models.values()]
            rows.append(dict(batch=i, loss=float(loss), n_mirror=int((m != 0).sum()),
                             coarse_max=float(g[0].abs().max()), fine_max=float(g[1].abs().max()),
                             coarse_norm=float(g[0].norm()), fine_norm=float(g[1].norm())))
    fm = np.array([r["fine_norm"] for r in rows])
    cm = np.array([r["coarse_norm"] for r in rows])
    med_f, med_c = float(np.median(fm)), float(np.median(cm))
    order = np.argsort(-(fm / med_f + cm / med_c))
    out = dict(n_batches=n_batches, median_fine_norm=med_f, median_coarse_norm=med_c,
               quantiles_fine_norm_over_median={q: float(np.quantile(fm / med_f, q)) for q in (0.5, 0.9, 0.99, 1.0)},
               quantiles_coarse_norm_over_median={q: float(np.quantile(cm / med_c, q)) for q in (0.5, 0.9, 0.99, 1.0)},
               top=[dict(rows[j], fine_over_median=float(fm[j] / med_f), coarse_over_median=float(cm[j] / med_c)) for j in order[:8]],
               median_batch=rows[int(np.argsort(fm)[len(fm) // 2])])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
