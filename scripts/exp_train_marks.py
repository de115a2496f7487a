#!/usr/bin/env python3
"""Round 5 tuning experiment: per-phase cycle counts (s_memtime marks, mean over workgroups) inside the TRAINING forward kernel
h2x::field_split_kernel<false,true,true,false> (positions, encoding, trunk, heads, density-gradient pass; operand planes stored on
the way), at the shapes of the training step.  Needs a library built with -DMNRF_EXP_CYCLES (scripts/build_variant.sh marks
"-DMNRF_EXP_CYCLES"; MNRF_LIB=exp_libs/marks.so).  Prints the kernel's launch time beside the marks."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mirror_nerf_amd as M  # noqa: E402,F401
from mirror_nerf_amd import _lib, synthetic as SY, training as T  # noqa: E402
from mirror_nerf_amd.weights import packed_of  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
system = M.NeRFSystem(T.default_hparams()).to(dev)
with torch.no_grad():
    for m in system.models.values():
        m.sigma.weight.mul_(20.0)
        m.sigma.bias.fill_(1.0)
model = system.nerf_fine
N, S = int(os.environ.get("RAYS", "1024")), int(os.environ.get("SPR", "192"))
rays = SY.device_rays(800, 800, dev)[torch.randint(0, 640000, (N,), device=dev)].contiguous()
z = torch.sort(torch.rand(N, S, device=dev) * 7 + 0.05, 1)[0].contiguous()
de = M.Embedding(4)(rays[:, 3:6].contiguous())
packed = packed_of(model)
L, p = _lib.lib(), _lib.ptr
B = N * S
f = lambda *s: torch.empty(*s, device=dev)  # noqa: E731
sig, rgb, pn, mir, nrm = f(B), f(B, 3), f(B, 3), f(B), f(B, 3)
save_x = torch.empty(L.mnrf_train_planes_bytes(B), dtype=torch.uint8, device=dev)
save_mask = torch.empty(L.mnrf_train_mask_words(B), dtype=torch.int64, device=dev)
save_inv, save_invj = f(B), f(B)
blocks = (B + 127) // 128
marks = torch.zeros(blocks * 16, dtype=torch.int64, device=dev)
has_marks = hasattr(L, "mnrf_exp_set_marks")
if has_marks:
    L.mnrf_exp_set_marks.argtypes = [ctypes.c_void_p]
    L.mnrf_exp_set_marks(marks.data_ptr())


def launch():
    _lib.check(L.mnrf_field_forward_train(p(packed), B, None, 3, p(rays), p(z), S, p(de), 27, p(sig), p(rgb), p(pn), p(mir), p(nrm),
                                          p(save_x), p(save_mask), p(save_inv), p(save_invj), _lib.MNRF_SPLIT_F16 | _lib.MNRF_TRAIN_PLANES,
                                          _lib.stream()), "fwd")


for _ in range(3):
    launch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    launch()
e1.record()
torch.cuda.synchronize()
print(f"{N} rays x {S}: {blocks} workgroups, {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per launch  (lib {os.environ.get('MNRF_LIB', 'default')})")
if has_marks:
    t = marks.view(blocks, 16).double()
    names = {1: "prologue", 2: "L1", 3: "L2-4", 4: "L5", 5: "L6-8", 6: "sigma", 7: "normal", 8: "mirror", 9: "final+view", 10: "dir+rgb",
             11: "grad pass", 15: "jacobian+close"}
    order = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 15]
    tot = float((t[:, 15] - t[:, 0]).mean())
    print(f"tile total {tot:10.0f} cycles (s_memtime, 100 MHz ticks x ... see DIARY 9.3)")
    for a, b in zip(order[:-1], order[1:]):
        d = float((t[:, b] - t[:, a]).mean())
        print(f"  {names[b]:16s} {d:10.0f}  {100 * d / tot:5.1f} %")
