"""Host time inside the custom autograd backward functions of one training step (they run on autograd's device thread, which
cProfile does not see): wraps every Function.backward of mirror_nerf_amd.autograd / losses with a wall-clock timer."""
import os
import sys
import time
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mirror_nerf_amd import autograd as A, losses as Ls, training  # noqa: E402
from oracle import mirror_nerf_oracle as O  # noqa: E402

acc = defaultdict(lambda: [0, 0.0])


def wrap(cls):
    orig = cls.backward

    def timed(ctx, *g):
        t = time.perf_counter()
        out = orig(ctx, *g)
        a = acc[cls.__name__]
        a[0] += 1
        a[1] += time.perf_counter() - t
        return out
    cls.backward = staticmethod(timed)


for mod in (A, Ls):
    for name in dir(mod):
        c = getattr(mod, name)
        if isinstance(c, type) and issubclass(c, torch.autograd.Function) and c is not torch.autograd.Function:
            wrap(c)
dev = torch.device("cuda", 0)
rays = torch.from_numpy(O.synthetic_rays(800, 800)).to(dev)
training.synthetic_train_bench(dev, rays, 3, 3, 1024)
acc.clear()
r = training.synthetic_train_bench(dev, rays, 20, 3, 1024)
print(r["ms_per_step"], "ms per step")
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:24s} {n:4d} calls  {t / n * 1e6:8.1f} us per call  {t / 23 * 1e6:8.1f} us per step")
