#!/usr/bin/env python3
"""Round 5 experiment: what the MLP launch of the hash-grid field (mf::tcnn_mfma_kernel<.,true>, behind the level-major encoding) is
bound by.  Time of one full evaluation of a 32768-ray chunk's fine pass through mnrf_tcnn_forward with an encoding workspace, minus
the encoding launch (mnrf_tcnn_encode), for libraries built with -DMNRF_EXP_TCNN_NO_PLANE_LOADS / -DMNRF_EXP_TCNN_NO_STORES
(scripts/build_variant.sh): the launch without its input traffic, without its output traffic."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mirror_nerf_amd as M  # noqa: E402
from mirror_nerf_amd import mirror_nerf as MN, synthetic as SY  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
models = {k: M.MirrorNeRFTcnn(encoding="hashgrid", bound=6.0, predict_normal=True, predict_mirror_mask=True).to(dev) for k in ("coarse", "fine")}
emb = {"xyz": M.Embedding(0), "dir": M.Embedding(0)}
rays = SY.device_rays(800, 800, dev)[300 * 800:300 * 800 + 32768].contiguous()
with torch.no_grad():
    rc = M.render_rays(models, emb, rays, 64, False, 0, 0, 128, 32768, test_time=True, compute_normal=False)
zf = rc["z_vals_fine"].contiguous()
m = models["fine"]
for f16 in (False, True):
    m.mlp_f16 = f16
    for sigma_only in (False, True):
        def run():
            with torch.no_grad():
                m.field(zf.numel(), rays=rays, z_vals=zf, spr=zf.shape[1], sigma_only=sigma_only)
        run()
        torch.cuda.synchronize()
        MN.LAUNCH_LOG = []
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for (_f, _B, a, b) in MN.LAUNCH_LOG) / 5
        MN.LAUNCH_LOG = None
        print(f"field() {'f16 MLPs' if f16 else 'hi/lo MLPs'} {'sigma only' if sigma_only else 'full'}: {ms:.3f} ms per {zf.numel()} samples "
              f"(encoding launch + MLP launch)  (lib {os.environ.get('MNRF_LIB', 'default')})")
