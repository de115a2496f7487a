"""Round 5 probe: does torch.cuda.graph (hipGraph) capture a training step whose kernels are launched through ctypes on
torch's current stream, with the backward pass driven by the autograd engine?  Primary rays only (render_rays, no recursion:
no host read anywhere in the step).  Prints eager vs replay time per step and the largest weight difference after 5 steps."""
import sys
import time

import torch

sys.path.insert(0, ".")
import mirror_nerf_amd as M  # noqa: E402
from mirror_nerf_amd import training as T  # noqa: E402
from mirror_nerf_amd.weights import params_of  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)


def make():
    torch.manual_seed(0)
    system = M.NeRFSystem(T.default_hparams()).to(dev)
    with torch.no_grad():
        for m in system.models.values():
            m.sigma.weight.mul_(20.0)
            m.sigma.bias.fill_(1.0)
    opt = T.FlatAdam(list(system.models.values()), lr=5e-4)
    return system, opt


N = 1024
rays = torch.from_numpy(__import__("oracle.mirror_nerf_oracle", fromlist=["x"]).synthetic_rays(100, 100)[:N].copy()).to(dev)
target = torch.rand(N, 3, device=dev)
gt = (torch.rand(N, device=dev) < 0.25).float()


def step(system, opt):
    hp = system.hparams
    res = M.render_rays(system.models, system.embeddings, rays, hp.N_samples, hp.use_disp, 0, 0, hp.N_importance, hp.chunk,
                        False, compute_normal=True, _guard=False)
    loss = T.color_mask_loss(res, target, gt)
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss


sys_a, opt_a = make()
for _ in range(3):
    step(sys_a, opt_a)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step(sys_a, opt_a)
torch.cuda.synchronize()
print("eager ms/step", (time.perf_counter() - t0) / 20 * 1e3)

sys_b, opt_b = make()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step(sys_b, opt_b)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        loss_static = step(sys_b, opt_b)
except Exception as e:  # noqa: BLE001
    print("CAPTURE FAILED:", type(e).__name__, e)
    raise
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
print("graph ms/step", (time.perf_counter() - t0) / 20 * 1e3, "loss", float(loss_static))
# the graph re-packs the weights inside (packed_of saw a stale generation at capture time)?  The loss must move
l0 = float(loss_static)
for _ in range(50):
    g.replay()
torch.cuda.synchronize()
print("loss after 50 more replays", float(loss_static), "(moved:", abs(float(loss_static) - l0) > 1e-6, ")")
