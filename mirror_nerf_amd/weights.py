"""Packing of a MirrorNeRF state dict into the MFMA-fragment image the field kernel streams.

The interchange format is the reference's parameter naming (`xyz_encoding_{1..8}.0.weight`,
`xyz_encoding_final`, `dir_encoding.0`, `sigma`, `rgb.0`, `normal_net.{0,1}`,
`is_mirror_net.{0,2}`; utils/__init__.py:109-136, train.py:56,66), so checkpoints of the
reference load unchanged.  Weights are read at call time: the packed image is rebuilt
whenever a parameter changed (optimizer steps bump `_version`).
"""
import ctypes

import torch

from . import _lib

PARAM_NAMES = []
for _i in range(8):
    PARAM_NAMES += [f"xyz_encoding_{_i+1}.0.weight", f"xyz_encoding_{_i+1}.0.bias"]
PARAM_NAMES += ["xyz_encoding_final.weight", "xyz_encoding_final.bias",
                "dir_encoding.0.weight", "dir_encoding.0.bias",
                "sigma.weight", "sigma.bias", "rgb.0.weight", "rgb.0.bias",
                "normal_net.0.weight", "normal_net.0.bias", "normal_net.1.weight", "normal_net.1.bias",
                "is_mirror_net.0.weight", "is_mirror_net.0.bias", "is_mirror_net.2.weight", "is_mirror_net.2.bias"]

PARAM_SHAPES = {}
for _i in range(8):
    _in = 63 if _i == 0 else (319 if _i == 4 else 256)
    PARAM_SHAPES[f"xyz_encoding_{_i+1}.0.weight"] = (256, _in)
    PARAM_SHAPES[f"xyz_encoding_{_i+1}.0.bias"] = (256,)
PARAM_SHAPES.update({
    "xyz_encoding_final.weight": (256, 256), "xyz_encoding_final.bias": (256,),
    "dir_encoding.0.weight": (128, 283), "dir_encoding.0.bias": (128,),
    "sigma.weight": (1, 256), "sigma.bias": (1,), "rgb.0.weight": (3, 128), "rgb.0.bias": (3,),
    "normal_net.0.weight": (128, 256), "normal_net.0.bias": (128,),
    "normal_net.1.weight": (3, 128), "normal_net.1.bias": (3,),
    "is_mirror_net.0.weight": (128, 256), "is_mirror_net.0.bias": (128,),
    "is_mirror_net.2.weight": (1, 128), "is_mirror_net.2.bias": (1,),
})


OPTIONAL_HEADS = ("normal_net.", "is_mirror_net.")     # predict_normal / predict_mirror_mask (mirror_nerf.py:80-99)
_ZEROS = {}


def _zeros(name, dev):
    """Stand-in weights of an absent optional head: the kernel evaluates the head on zeros, nobody reads its output."""
    t = _ZEROS.get((name, str(dev)))
    if t is None:
        t = torch.zeros(PARAM_SHAPES[name], dtype=torch.float32, device=dev)
        _ZEROS[(name, str(dev))] = t
    return t


def canonical(name, t):
    """A parameter of a model with fewer encoding bands (--N_emb_xyz < 10, --N_emb_dir < 4; opt.py:35-46) in the shape the
    kernels are built for: the columns of the absent bands as zeros.  Band k of Embedding occupies channels 3 + 6 k .. 8 + 6 k
    (models/mirror_nerf.py:30-38), so the channels of a shorter encoding are a PREFIX of the 63 / 27."""
    want = PARAM_SHAPES[name]
    if tuple(t.shape) == want or t.dim() != 2 or t.shape[0] != want[0] or t.shape[1] > want[1]:
        return t
    z = lambda n: torch.zeros(t.shape[0], n, dtype=t.dtype, device=t.device)  # noqa: E731
    if name == "xyz_encoding_5.0.weight":                  # cat([xyz encoding, h]) (mirror_nerf.py:192-193)
        c = t.shape[1] - 256
        return torch.cat([t[:, :c], z(63 - c), t[:, c:]], 1) if 3 <= c <= 63 else t
    if name in ("xyz_encoding_1.0.weight", "dir_encoding.0.weight"):      # the encoding is the (only / last) block of columns
        return torch.cat([t, z(want[1] - t.shape[1])], 1)
    return t


def decanonical(name, g, shape):
    """The gradient of a canonical-shape parameter cut back to the model's own shape (inverse of `canonical`)."""
    if tuple(g.shape) == tuple(shape):
        return g
    if name == "xyz_encoding_5.0.weight":
        c = shape[1] - 256
        return torch.cat([g[:, :c], g[:, 63:]], 1)
    return g[:, :shape[1]]


def _param_pointers(tensors, arr, base, keep):
    """Fill arr[base : base + 32] with the addresses of one model's parameters in PARAM_NAMES order; returns the device."""
    missing = [n for n in PARAM_NAMES if n not in tensors and not n.startswith(OPTIONAL_HEADS)]
    if missing:
        raise RuntimeError(
            "the HIP field kernel implements the reference's D=8, W=256 trunk with the colour head "
            f"(train.py:44-66); missing parameters: {missing}")
    dev = next(iter(tensors.values())).device
    for i, n in enumerate(PARAM_NAMES):
        t = canonical(n, tensors[n].detach()) if n in tensors else _zeros(n, dev)
        if tuple(t.shape) != PARAM_SHAPES[n]:
            raise RuntimeError(f"{n}: shape {tuple(t.shape)} != {PARAM_SHAPES[n]} (D=8, W=256, N_emb 10/4 only)")
        if t.dtype != torch.float32 or not t.is_cuda:
            raise RuntimeError(f"{n}: need a float32 CUDA tensor")
        t = t.contiguous()
        keep.append(t)
        arr[base + i] = t.data_ptr()
        dev = t.device
    return dev


def pack_states(states, outs=None):
    """states: list of dicts name -> fp32 CUDA tensor (one per model, all on one device); outs: list of images to overwrite (entries
    may be None).  ONE mnrf_pack_weights_n call for all of them.  Returns the packed images (1-D fp32 CUDA tensors).
    The parameters of an absent optional head (predict_normal=False / predict_mirror_mask=False) are packed as zeros."""
    L = _lib.lib()
    n = len(states)
    outs = list(outs) if outs is not None else [None] * n
    keep = []
    arr = (ctypes.c_void_p * (_lib.N_PARAMS * n))()
    img = (ctypes.c_void_p * n)()
    for m, tensors in enumerate(states):
        dev = _param_pointers(tensors, arr, m * _lib.N_PARAMS, keep)
        if outs[m] is None:
            outs[m] = torch.empty(L.mnrf_packed_floats(), dtype=torch.float32, device=dev)
        img[m] = outs[m].data_ptr()
    _lib.check(L.mnrf_pack_weights_n(n, arr, img, _lib.stream()), "mnrf_pack_weights")
    return outs


def pack_state(tensors, out=None):
    """One model (see pack_states)."""
    return pack_states([tensors], [out])[0]


def param_refs(module):
    """[(submodule, local name, qualified name)] of a module's parameters in named_parameters() order, cached on the module:
    walking the module tree costs ~0.25 ms per call and the training step needs the list eight times.  The Parameter
    objects are looked up at every use (so `.to()`, load_state_dict and re-assigned parameters are seen); only adding or
    removing submodules after the first call would go unnoticed."""
    refs = module.__dict__.get("_mnrf_param_refs")
    if refs is None:
        refs = [(sub, pname, (mname + "." if mname else "") + pname)
                for mname, sub in module.named_modules() for pname, q in sub._parameters.items() if q is not None]
        module.__dict__["_mnrf_param_refs"] = refs
    return refs


def params_of(module):
    return [sub._parameters[pname] for sub, pname, _ in param_refs(module)]


# Optimizer steps must invalidate every packed image.  `Tensor._version` is NOT enough: torch's fused Adam/AdamW/SGD
# (`fused=True`, one multi-tensor kernel) update the parameters without bumping it (measured on torch 2.10: the version of
# every parameter stays where it was, the packed image went stale and training stood still while the loss only jittered
# with the batches).  A global post-step hook on every torch optimizer bumps a generation number that is part of the
# cache key; inference never steps an optimizer, so nothing is re-packed there.
_GENERATION = [0]


def bump_generation(*_a, **_k):
    _GENERATION[0] += 1


try:
    from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_hook
    _reg_hook(bump_generation)
except ImportError:      # very old torch: fall back to re-packing whenever a parameter requires grad
    _reg_hook = None


class PackedCache:
    """Packed image of one nn.Module (ours or the reference's), refreshed when parameters change.

    "Changed" = a parameter was re-assigned, moved, written through an autograd-visible in-place op (`load_state_dict`,
    `p.mul_()` under no_grad ...: those bump `Tensor._version`), or ANY torch optimizer took a step (global post-step hook,
    see _GENERATION).  Writes through `p.data` outside an optimizer (some EMA / clamping code does that) are invisible to
    both -- call `invalidate_packed(module)` after such surgery."""

    def __init__(self):
        self.key = None
        self.packed = None

    def stale(self, module):
        """-> the key the image should have, or None when the image is current."""
        # (the look-up runs eight times per training step, four of them with nothing else queued on the device: one pass over
        # the parameters, no dict unless the image has to be rebuilt)
        refs = param_refs(module)
        key = [_GENERATION[0]]
        for sub, pname, _ in refs:
            q = sub._parameters[pname]
            key.append(q.data_ptr())
            key.append(q._version)
        if _reg_hook is None and any(sub._parameters[pname].requires_grad for sub, pname, _ in refs):
            self.key = None
        return key if (key != self.key or self.packed is None) else None

    def get(self, module):
        key = self.stale(module)
        if key is not None:
            self.packed = pack_state({full: sub._parameters[pname] for sub, pname, full in param_refs(module)}, self.packed)
            self.key = key
        return self.packed


def packed_of_many(modules):
    """packed_of for several MirrorNeRF modules; the stale images among them are rebuilt by ONE batched launch pair."""
    caches, todo = [], []
    for m in modules:
        cache = m.__dict__.get("_mnrf_packed")
        if cache is None:
            cache = m.__dict__["_mnrf_packed"] = PackedCache()
        caches.append(cache)
        key = cache.stale(m)
        if key is not None:
            todo.append((m, cache, key))
    if len({next(iter(m.parameters())).device for m, _, _ in todo}) > 1:      # (images on several devices: one call each)
        return [c.get(m) for m, c in zip(modules, caches)]
    if todo:
        outs = pack_states([{full: sub._parameters[pname] for sub, pname, full in param_refs(m)} for m, _, _ in todo],
                           [c.packed for _, c, _ in todo])
        for (m, c, key), out in zip(todo, outs):
            c.packed, c.key = out, key
    return [c.packed for c in caches]


def invalidate_packed(module):
    """Force the next launch to re-pack `module`'s weights (needed only after writes through `param.data`, which
    bypass the version counter the cache is keyed on).  Covers the MirrorNeRF packed image and the hash-grid field's
    weight blob."""
    cache = module.__dict__.get("_mnrf_packed")
    if cache is not None:
        cache.key = None
    if hasattr(module, "_blob_key"):
        module._blob_key = None


# Inside ONE forward call of NeRFSystem / batched_inference nobody but this package touches the weights: the images are validated
# once at the start (`with validated(models):`) and every look-up in that dynamic extent is a dict hit instead of 64 property
# reads (35 us each time, twice per step with nothing else queued on the device).
_VALIDATED = {}


class validated:
    def __init__(self, modules):
        self.modules = [m for m in modules if hasattr(m, "parameters")]

    def __enter__(self):
        self.added = []
        fresh = [m for m in self.modules if id(m) not in _VALIDATED and m.__class__.__name__ == "MirrorNeRF"]
        for m, image in zip(fresh, packed_of_many(fresh) if fresh else []):
            _VALIDATED[id(m)] = image
            self.added.append(id(m))
        return self

    def __exit__(self, *exc):
        for k in self.added:
            _VALIDATED.pop(k, None)
        return False


def packed_of(module):
    hit = _VALIDATED.get(id(module))
    if hit is not None:
        return hit
    cache = module.__dict__.get("_mnrf_packed")
    if cache is None:
        cache = PackedCache()
        module.__dict__["_mnrf_packed"] = cache
    return cache.get(module)
