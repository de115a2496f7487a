"""One 800x800 frame of the hash-grid model (config 5, eval recursion) twice; under rocprofv3 --kernel-trace the second frame's
busy / idle time shows whether the frame is device- or host-bound.  python scripts/trace_tcnn_frame.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mirror_nerf_amd as M  # noqa: E402
from oracle import mirror_nerf_oracle as O  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
models = {k: M.MirrorNeRFTcnn(encoding="hashgrid", bound=6.0, predict_normal=True, predict_mirror_mask=True).to(dev) for k in ("coarse", "fine")}
emb = {"xyz": M.Embedding(0), "dir": M.Embedding(0)}
rays = torch.from_numpy(O.synthetic_rays(800, 800)).to(dev)
args = dict(predict_normal=True, only_one_field=False, only_one_field_fine_epoch=2, max_recursive_level=1,
            app_control_mirror_roughness=False, trace_ray_times=1)


def frame():
    return M.batched_inference(models, emb, rays, 64, 128, False, 32768, args=args, trace_secondary_rays=True, normal_noise_std=0.0,
                               test_time=True, white_back=False, to_cpu=False)


frame()
torch.cuda.synchronize()
t0 = time.perf_counter()
frame()
torch.cuda.synchronize()
print(f"frame {1e3 * (time.perf_counter() - t0):.1f} ms")
