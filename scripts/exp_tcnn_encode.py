#!/usr/bin/env python3
"""Round 5 experiment: time of the level-major encoding launch (mnrf_tcnn_encode) on one 32768-ray chunk of the bench frame with its
fine-pass depths, for libraries built with different thread -> sample maps (-DMNRF_EXP_ENC_PATCH_RAYS=R)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mirror_nerf_amd as M  # noqa: E402
from mirror_nerf_amd import _lib, synthetic as SY  # noqa: E402
from mirror_nerf_amd.mirror_nerf_tcnn import _offsets17  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
models = {k: M.MirrorNeRFTcnn(encoding="hashgrid", bound=6.0, predict_normal=True, predict_mirror_mask=True).to(dev) for k in ("coarse", "fine")}
emb = {"xyz": M.Embedding(0), "dir": M.Embedding(0)}
rays = SY.device_rays(800, 800, dev)[300 * 800:300 * 800 + 32768].contiguous()
with torch.no_grad():
    rc = M.render_rays(models, emb, rays, 64, False, 0, 0, 128, 32768, test_time=True, compute_normal=False)
zf = rc["z_vals_fine"].contiguous()
m = models["fine"]
table = m.encoder.embeddings.detach().contiguous()
offs = _offsets17(m.cfg)
planes = torch.empty(32 * zf.numel(), device=dev)
ref = None


def pr():
    _lib.check(_lib.lib().mnrf_tcnn_encode(_lib.ptr(table), offs, m.cfg["S"], m.cfg["H"], float(m.bound), zf.numel(), None, 0, _lib.ptr(rays),
                                           _lib.ptr(zf), zf.shape[1], _lib.ptr(planes), _lib.stream()), "encode")


pr()
torch.cuda.synchronize()
chk = float(planes.double().sum()), float(planes.double().abs().sum())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    pr()
e1.record()
torch.cuda.synchronize()
bits = int(planes.view(torch.int32).to(torch.int64).sum())      # exact: equal for bit-identical planes
# the same launch from the half2 copy of the table
th, tflag = m._table() if hasattr(m, "_table") else (None, 0)
m.table_f16 = True
th, tflag = m._table()
m.table_f16 = False


def pr16():
    _lib.check(_lib.lib().mnrf_tcnn_encode_flags(th.data_ptr(), offs, m.cfg["S"], m.cfg["H"], float(m.bound), zf.numel(), None, 0, _lib.ptr(rays),
                                                 _lib.ptr(zf), zf.shape[1], _lib.ptr(planes), tflag, _lib.stream()), "encode (half2)")


pr16()
torch.cuda.synchronize()
bits16 = int(planes.view(torch.int32).to(torch.int64).sum())
h0, h1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
h0.record()
for _ in range(10):
    pr16()
h1.record()
torch.cuda.synchronize()
print(f"half2 table: {h0.elapsed_time(h1) / 10:.3f} ms per launch, bits {bits16}; fp32 table bits {bits}")
print(f"encode {e0.elapsed_time(e1) / 10:.3f} ms per launch ({zf.numel()} samples), checksum {chk[0]:.6e} {chk[1]:.6e}  (lib {os.environ.get('MNRF_LIB', 'default')})")
