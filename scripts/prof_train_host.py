"""Host-side profile of the training step (cProfile): where the Python/launch time of a step goes.
python scripts/prof_train_host.py [tottime|cumtime]   (cumtime: this package's functions only)"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mirror_nerf_amd import training  # noqa: E402
from oracle import mirror_nerf_oracle as O  # noqa: E402

dev = torch.device("cuda", 0)
rays = torch.from_numpy(O.synthetic_rays(800, 800)).to(dev)
training.synthetic_train_bench(dev, rays, 3, 3, 1024)
pr = cProfile.Profile()
pr.enable()
r = training.synthetic_train_bench(dev, rays, 20, 3, 1024)
pr.disable()
print(r["ms_per_step"], "ms per step under the profiler; 23 steps profiled")
st = pstats.Stats(pr)
if len(sys.argv) > 1 and sys.argv[1] == "cumtime":
    st.sort_stats("cumtime").print_stats("mirror_nerf_amd", 45)
else:
    st.sort_stats("tottime").print_stats(28)
