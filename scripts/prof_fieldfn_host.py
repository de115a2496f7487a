"""Where the host time of FieldFn.backward goes: line-level timing with sys.setprofile-free checkpoints (monkey-patched copies
of the hot helpers)."""
import os
import sys
import time
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mirror_nerf_amd import autograd as A, training, _lib  # noqa: E402
from oracle import mirror_nerf_oracle as O  # noqa: E402

acc = defaultdict(lambda: [0, 0.0])


def timed(owner, name, label=None):
    orig = getattr(owner, name)

    def f(*a, **k):
        t = time.perf_counter()
        out = orig(*a, **k)
        e = acc[label or name]
        e[0] += 1
        e[1] += time.perf_counter() - t
        return out
    setattr(owner, name, f)


timed(A._Pending, "__init__", "_Pending.__init__")
timed(A._Pending, "all32", "_Pending.all32")
timed(A._Pending, "finish", "_Pending.finish")
L = _lib.lib()
for fn in ("mnrf_field_backward_planes", "mnrf_dw_planes2", "mnrf_ray_grads", "mnrf_dw_planes2_workspace_floats"):
    orig = getattr(L, fn)

    def mk(orig=orig, fn=fn):
        def f(*a):
            t = time.perf_counter()
            out = orig(*a)
            e = acc["C " + fn]
            e[0] += 1
            e[1] += time.perf_counter() - t
            return out
        return f
    setattr(L, fn, mk())
from mirror_nerf_amd import weights as W  # noqa: E402
timed(W, "packed_of")
timed(W, "decanonical")
orig_b = A.FieldFn.backward


def tb(ctx, *g):
    t = time.perf_counter()
    out = orig_b(ctx, *g)
    e = acc["FieldFn.backward total"]
    e[0] += 1
    e[1] += time.perf_counter() - t
    return out


A.FieldFn.backward = staticmethod(tb)
dev = torch.device("cuda", 0)
rays = torch.from_numpy(O.synthetic_rays(800, 800)).to(dev)
training.synthetic_train_bench(dev, rays, 3, 3, 1024)
acc.clear()
r = training.synthetic_train_bench(dev, rays, 20, 3, 1024)
print(r["ms_per_step"], "ms per step")
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:40s} {n:5d} calls  {t / n * 1e6:8.1f} us per call  {t / 23 * 1e6:8.1f} us per step")
