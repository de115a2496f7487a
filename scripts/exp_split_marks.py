#!/usr/bin/env python3
"""Tuning experiment: per-phase cycle counts (s_memtime marks, averaged over workgroups) inside the split-f16 field kernels.
Needs a library built with -DMNRF_EXP_CYCLES on mnrf_field_split.hip / mnrf_field_split32.hip (MNRF_LIB=...); MNRF_SPLIT32
selects the tuning as usual."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import mirror_nerf_amd as M  # noqa: E402,F401
from mirror_nerf_amd import _lib  # noqa: E402
from mirror_nerf_amd.weights import packed_of  # noqa: E402
from oracle import mirror_nerf_oracle as O  # noqa: E402

dev = torch.device("cuda", 0)
models, sds, emb = bench.build_models(dev)
rays = torch.from_numpy(O.synthetic_rays(800, 800)[300 * 800:300 * 800 + 32768]).to(dev)
S = 192
z = torch.sort(torch.rand(32768, S, device=dev) * 7 + 0.05, 1)[0].contiguous()
dir_emb = emb["dir"](rays[:, 3:6].contiguous())
packed = packed_of(models["fine"])
B = 32768 * S
f = lambda *s: torch.empty(*s, device=dev)  # noqa: E731
sig, rgb, pn, mir = f(B), f(B, 3), f(B, 3), f(B)
blocks = B // 128
dbg = torch.zeros(blocks * 16, dtype=torch.int64, device=dev)
p = _lib.ptr
for it in range(3):
    _lib.check(_lib.lib().mnrf_field_forward(p(packed), _lib.MNRF_SPLIT_F16, B, None, 3, p(rays), p(z), S, p(dir_emb), 27, p(sig),
                                             p(rgb), p(pn), p(mir), None, p(dbg), _lib.stream()), "f")
torch.cuda.synchronize()
t = dbg.view(blocks, 16)[2048:40000].double()
names = {1: "prologue", 2: "L1", 3: "L2-4", 4: "L5", 5: "L6-8", 6: "sigma", 7: "normal", 8: "mirror", 9: "final", 10: "dir+rgb", 15: "close"}
order = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 15]
print("tile total", float((t[:, 15] - t[:, 0]).mean()))
for a, b in zip(order[:-1], order[1:]):
    print(f"{names[b]:10s} {float((t[:, b] - t[:, a]).mean()):10.0f}")
