"""Second-order weight gradients: rows route (mnrf_field_backward2) and planes route (mnrf_field_backward2_planes +
mnrf_dw_planes2) against torch's double backward in FLOAT64 on the same samples.  Prints each route's error relative to the
tensor's largest entry.  GPU box: python scripts/check_so_routes.py [B]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import mirror_nerf_amd as M  # noqa: E402
from mirror_nerf_amd import _lib  # noqa: E402
from mirror_nerf_amd.weights import PARAM_NAMES, PARAM_SHAPES, packed_of  # noqa: E402
import torch_ref as TR  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 200
DEV = "cuda:0"
L, p = _lib.lib(), _lib.ptr
f = lambda *s: torch.empty(*s, dtype=torch.float32, device=DEV)  # noqa: E731
torch.manual_seed(B + 5)
model = M.MirrorNeRF(in_channels_xyz=63, in_channels_dir=27, predict_normal=True, predict_mirror_mask=True).to(DEV)
with torch.no_grad():
    model.sigma.weight.mul_(20.0)
packed = packed_of(model)
xyz = (torch.rand(B, 3, device=DEV) * 6 - 3).contiguous()
de = M.Embedding(4)(torch.nn.functional.normalize(torch.randn(B, 3, device=DEV), dim=1))
o = (f(B), f(B, 3), f(B, 3), f(B), f(B, 3))
sx = torch.zeros(L.mnrf_train_planes_bytes(B), dtype=torch.uint8, device=DEV)
sm = torch.zeros(L.mnrf_train_mask_words(B), dtype=torch.int64, device=DEV)
si, sj = f(B), f(B)
_lib.check(L.mnrf_field_forward_train(p(packed), B, p(xyz), 3, None, None, 1, p(de), 27, *[p(t) for t in o], p(sx), p(sm),
                                      p(si), p(sj), _lib.MNRF_SPLIT_F16 | _lib.MNRF_TRAIN_PLANES, _lib.stream()), "forward")
normal = o[4]
flat = os.environ.get("FLAT", "0") == "1"
scale = torch.ones(B, device=DEV) if flat else 10.0 ** (torch.rand(B, device=DEV) * 8 - 8)
g_n = (torch.randn(B, 3, device=DEV) * scale[:, None]).contiguous()
arr = lambda ts: (ctypes.c_void_p * 32)(*[t.data_ptr() for t in ts])  # noqa: E731
d_r = [torch.zeros(*PARAM_SHAPES[n], device=DEV) for n in PARAM_NAMES]
ws2 = f(L.mnrf_train_workspace2_floats(B))
_lib.check(L.mnrf_field_backward2(p(packed), B, p(xyz), 3, None, None, 1, p(g_n), p(normal), p(sj), p(sm), p(ws2),
                                  arr(d_r), None, _lib.MNRF_SPLIT_F16, _lib.stream()), "rows")
x2 = torch.zeros(L.mnrf_train_planes2_bytes(B), dtype=torch.uint8, device=DEV)
y2 = torch.zeros(L.mnrf_train_dy_planes2_bytes(B), dtype=torch.uint8, device=DEV)
jmax = torch.zeros(1, dtype=torch.int32, device=DEV)
_lib.check(L.mnrf_field_backward2_planes(p(packed), B, p(xyz), 3, None, None, 1, p(g_n), p(normal), p(sj), p(sm),
                                         p(x2), p(y2), p(jmax), None, _lib.stream()), "planes")
d_p = [torch.zeros(*PARAM_SHAPES[n], device=DEV) for n in PARAM_NAMES]
bs = (ctypes.c_int64 * 1)(B)
kd = (ctypes.c_int * 1)(1)
wsp = f(L.mnrf_dw_planes2_workspace_floats(1, bs, kd))
_lib.check(L.mnrf_dw_planes2(1, (ctypes.c_void_p * 1)(x2.data_ptr()), (ctypes.c_void_p * 1)(y2.data_ptr()), bs,
                             (ctypes.c_void_p * 1)(jmax.data_ptr()), kd, p(wsp), arr(d_p), 0, _lib.stream()), "dw planes2")
torch.cuda.synchronize()
# float64 truth: d/dW of sum(normal . g_n) with normal = l2n(-d sigma/d x) built with create_graph
w = {k: v.detach().double().cpu().requires_grad_(True) for k, v in model.state_dict().items()}
x64 = xyz.double().cpu().requires_grad_(True)
outs = TR.field(w, x64, de.double().cpu(), with_normal=True)
(outs[4] * g_n.double().cpu()).sum().backward()
rel = lambda a, b: float((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-300))  # noqa: E731
print(f"B = {B}, g_normal {'flat' if flat else 'over 8 decades'}; error / largest entry:   rows      planes")
for n, a, b in zip(PARAM_NAMES, d_r, d_p):
    t = w[n].grad
    if t is None or float(t.abs().max()) == 0.0:
        continue
    print(f"  {n:28s} {rel(a, t):.2e}  {rel(b, t):.2e}")
