"""Deterministic MirrorNeRF weights for fixtures, tests and the benchmark.

The golden fixtures do not store the 2.65 MB state dict of each model; they
store the seed, the list of tweaks and a checksum.  `make_state_dict` rebuilds
the weights with plain `torch.nn.Linear` default initialisation in the same
construction order as the reference constructor (models/mirror_nerf.py:41-99),
which `make_golden.py` verifies bit-for-bit against the reference module built
under the same seed.  torch's CPU generator is deterministic across machines
for a given torch build; `checksum` makes a silent mismatch impossible.
"""
import numpy as np
import torch
from torch import nn


def make_state_dict(seed, n_models=1, W=256, D=8, in_xyz=63, in_dir=27, skips=(4,),
                    predict_normal=True, predict_mirror_mask=True):
    """Returns a list of `n_models` dicts name -> float32 ndarray (state_dict names).

    Models are constructed one after the other after a single manual_seed, like
    train.py:44-66 builds nerf_coarse then nerf_fine."""
    torch.manual_seed(seed)
    out = []
    for _ in range(n_models):
        sd = {}

        def lin(name, i, o):
            m = nn.Linear(i, o)
            sd[name + ".weight"] = m.weight.detach().numpy().copy()
            sd[name + ".bias"] = m.bias.detach().numpy().copy()

        for i in range(D):
            if i == 0:
                lin(f"xyz_encoding_{i+1}.0", in_xyz, W)
            elif i in skips:
                lin(f"xyz_encoding_{i+1}.0", W + in_xyz, W)
            else:
                lin(f"xyz_encoding_{i+1}.0", W, W)
        lin("xyz_encoding_final", W, W)
        lin("dir_encoding.0", W + in_dir, W // 2)
        lin("sigma", W, 1)
        lin("rgb.0", W // 2, 3)
        if predict_normal:
            lin("normal_net.0", W, W // 2)
            lin("normal_net.1", W // 2, 3)
        if predict_mirror_mask:
            lin("is_mirror_net.0", W, W // 2)
            lin("is_mirror_net.2", W // 2, 1)
        out.append(sd)
    return out


def apply_tweaks(sd, tweaks):
    """tweaks: list of [name, op, value], op in {"mul", "set", "add"} -- in place."""
    for name, op, val in tweaks:
        a = sd[name]
        if op == "mul":
            a *= np.float32(val)
        elif op == "set":
            a[...] = np.float32(val)
        elif op == "add":
            a += np.float32(val)
        else:
            raise ValueError(op)
    return sd


def checksum(sd):
    """Order-independent, position-sensitive digest of a state dict (float64)."""
    tot = 0.0
    for name in sorted(sd):
        a = sd[name].astype(np.float64).ravel()
        tot += float(np.dot(a, np.cos(np.arange(a.size, dtype=np.float64) * 0.37 + 0.11)))
    return tot


# density tweak used by most fixtures so that rays become opaque (SURVEY 8a noise-floor probe)
OPAQUE = [["sigma.weight", "mul", 20.0], ["sigma.bias", "set", 1.0]]
# mirror head straddles 0.5 (fixture G6/G7)
# (about 40 % of the rays above 0.5 at recursion levels 0 and 1, none closer than 2e-4 to 0.5)
STRADDLE = [["is_mirror_net.2.weight", "mul", 200.0], ["is_mirror_net.2.bias", "add", -0.38],
            ["sigma.bias", "set", 5.0], ["sigma.weight", "mul", 20.0]]
# every ray is a mirror (fixture G8)
ALL_MIRROR = OPAQUE + [["is_mirror_net.2.bias", "set", 10.0]]
