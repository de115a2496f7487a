"""mnrf_adam_step against torch.optim.Adam(fused=True) on one flat tensor: gradients over 30 orders of magnitude, 5 steps."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mirror_nerf_amd import _lib  # noqa: E402

dev = "cuda:0"
L, p = _lib.lib(), _lib.ptr
torch.manual_seed(0)
n = 595_333
p0 = torch.randn(n, device=dev) * 0.1
a = torch.nn.Parameter(p0.clone())
opt = torch.optim.Adam([a], lr=5e-4, fused=True)
b = p0.clone()
m, v = torch.zeros_like(b), torch.zeros_like(b)
skipped = torch.zeros(1, dtype=torch.int32, device=dev)
for step in range(1, 6):
    g = torch.randn(n, device=dev) * 10.0 ** (torch.rand(n, device=dev) * 30 - 28)
    g[::7] = 0.0
    a.grad = g.clone()
    opt.step()
    _lib.check(L.mnrf_adam_step(p(b), p(g), p(m), p(v), n, 5e-4, 0.9, 0.999, 1e-8, 0.0, step, p(skipped), None, None, _lib.stream()), "adam")
    d = (a.detach() - b).abs()
    print(step, "max |torch - kernel| =", float(d.max()), "at g =", float(g[d.argmax()]), " mean", float(d.mean()))
