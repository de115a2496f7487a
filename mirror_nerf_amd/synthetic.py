"""Deterministic MirrorNeRF weights and the synthetic camera for fixtures, tests, smoke() and the benchmark.

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
    """tweaks: list of [name, op, value], op in {"mul", "set", "add", "mul_cols"} -- in place.
    "mul_cols": value = [first column, end column, factor] (a band of input channels of a weight matrix)."""
    for name, op, val in tweaks:
        a = sd[name]
        if op == "mul_cols":
            a[:, int(val[0]):int(val[1])] *= np.float32(val[2])
        elif op == "mul":
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
# a rough spectrum on top of trained weights (fixtures G11 "rough"): the 2^7..2^9 encoding bands (input columns 45..62
# of the two layers that read the encoding; in layer 5 the encoding comes first, models/mirror_nerf.py:192-193) carry
# 6x their trained weight and the density head 25x -- sigma of ~1e3 at surfaces, the worst case SURVEY 8a names
ROUGH = [["xyz_encoding_1.0.weight", "mul_cols", [45, 63, 6.0]], ["xyz_encoding_5.0.weight", "mul_cols", [45, 63, 6.0]],
         ["sigma.weight", "mul", 25.0], ["sigma.bias", "mul", 25.0]]


# ----------------------------------------------------------------------------- synthetic camera
CAMERA_ANGLE_X = 0.6911112     # datasets/blender.py:40-42 reads it from transforms.json; the lego value
NEAR, FAR = 0.05, 8.0          # run.sh:14-15


def look_at_pose(eye=(0.0, -4.0, 1.5), target=(0.0, 0.0, 0.0), up=(0.0, 0.0, 1.0)):
    """Camera-to-world (3,4) fp32 of a camera at `eye` looking at `target` (OpenGL convention: the camera looks
    along -z, datasets/ray_utils.py:22-24)."""
    eye, target, up = (np.asarray(v, dtype=np.float64) for v in (eye, target, up))
    zc = eye - target
    zc /= np.linalg.norm(zc)
    xc = np.cross(up, zc)
    xc /= np.linalg.norm(xc)
    yc = np.cross(zc, xc)
    return np.stack([xc, yc, zc, eye], 1).astype(np.float32)


def device_rays(H, W, dev, pose=None, near=NEAR, far=FAR, camera_angle_x=CAMERA_ANGLE_X):
    """(H*W, 8) rays [o, d, near, far] of a pin-hole camera, generated on the GPU by mnrf_generate_rays
    (datasets/ray_utils.py:6-53, datasets/blender.py:159-168)."""
    import ctypes
    from . import _lib
    pose = look_at_pose() if pose is None else pose
    focal = 0.5 * W / np.tan(0.5 * camera_angle_x)
    rays = torch.empty(H * W, 8, device=dev)
    c2w = (ctypes.c_float * 12)(*np.asarray(pose, np.float32).reshape(-1).tolist())
    with torch.cuda.device(rays.device):
        _lib.check(_lib.lib().mnrf_generate_rays(H, W, float(focal), c2w, float(near), float(far), _lib.ptr(rays),
                                                 _lib.stream()), "mnrf_generate_rays")
    return rays


def build_models(dev, tweaks=None, seed=0, names=("coarse", "fine")):
    """The seeded random-init MirrorNeRF pair of the benchmark / smoke test on `dev` (+ its numpy state dicts)."""
    from .mirror_nerf import MirrorNeRF
    sds = [apply_tweaks(sd, tweaks or []) for sd in make_state_dict(seed, len(names))]
    models = {}
    for name, sd in zip(names, sds):
        m = MirrorNeRF(in_channels_xyz=63, in_channels_dir=27, predict_normal=True, predict_mirror_mask=True)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        models[name] = m.to(dev)
    return models, sds


def make_tcnn_table(n_entries, seed, scale, keep_levels=None, offsets=None):
    """Deterministic hash-grid table (n_entries, 2) fp32 for fixtures and tests: uniform(-scale, scale) from numpy's
    RandomState(seed); with `keep_levels` = k the levels k.. are zero (a random table is white noise at the fine levels:
    d(output)/d(position) ~ 1e4, which makes the comparison of two fp32 implementations ill-conditioned)."""
    t = np.random.RandomState(seed).uniform(-scale, scale, (int(n_entries), 2)).astype(np.float32)
    if keep_levels is not None:
        t[int(offsets[keep_levels]):] = 0
    return t
