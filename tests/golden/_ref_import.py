"""Import shim for the read-only reference tree (build container only).

The reference lives at /root/reference and is a Python program, so it can be
imported here to (a) validate the oracle and (b) generate golden vectors.  It
never travels to the GPU box: nothing under tests/ that is marked `gpu`,
`__graft_entry__.smoke()` or `bench.py` may import this module.

The reference needs a few third-party packages that are absent from this image
(pytorch_lightning, kornia, cv2, torchvision, torch_optimizer, imageio).  None
of them is on the arithmetic path that the fixtures pin, so they are replaced
by empty stand-in modules.  SURVEY.md section 8(c) lists the recipe.
"""
import os
import sys
import types

REF_ROOT = "/root/reference"


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "models"))


def _stub(name, **attrs):
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    sys.modules[name] = mod
    return mod


def _create_meshgrid(height, width, normalized_coordinates=True, device=None, dtype=None):
    """kornia.utils.create_meshgrid (kornia is absent from this image and un-pinned in the reference's README):
    its published definition -- xs = linspace(0, W-1, W), ys = linspace(0, H-1, H), grid[0, y, x] = (xs[x], ys[y]),
    shape (1, H, W, 2).  The reference only calls it with normalized_coordinates=False (datasets/ray_utils.py:18),
    where every entry is an exactly representable integer, so there is no rounding to pin."""
    import torch
    if normalized_coordinates:
        raise NotImplementedError("only the un-normalised grid is used by the reference's ray generator")
    xs = torch.linspace(0, width - 1, width, device=device, dtype=dtype)
    ys = torch.linspace(0, height - 1, height, device=device, dtype=dtype)
    return torch.stack(torch.meshgrid([xs, ys], indexing="ij"), dim=-1).permute(1, 0, 2).unsqueeze(0)


def ray_utils():
    """datasets/ray_utils.py imported by path (the `datasets` package itself pulls cv2/PIL dataset classes)."""
    install()
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_ray_utils", os.path.join(REF_ROOT, "datasets", "ray_utils.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def install():
    """Make `models.rendering`, `models.mirror_nerf`, `train`, `eval` importable."""
    if not available():
        raise RuntimeError("reference tree not present at " + REF_ROOT)
    sys.dont_write_bytecode = True  # never write __pycache__ into the reference
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)

    import torch
    from torch import nn

    # `utils/__init__.py` pulls torch_optimizer/cv2/torchvision; only utils.func
    # is on the hot path, so expose the package directory without running it.
    if "utils" not in sys.modules or not hasattr(sys.modules["utils"], "__path__"):
        pkg = types.ModuleType("utils")
        pkg.__path__ = [os.path.join(REF_ROOT, "utils")]
        sys.modules["utils"] = pkg
        # names that `from utils import *` in train.py is expected to provide
        pkg.load_ckpt = lambda model, path, name: None
        pkg.__all__ = ["load_ckpt"]

    class _LightningModule(nn.Module):
        def __init__(self):
            super().__init__()
            self.current_epoch = 0

        def save_hyperparameters(self, hparams):
            self.hparams = hparams

        def log(self, *a, **k):
            pass

    _stub(
        "pytorch_lightning",
        LightningModule=_LightningModule,
        Trainer=object,
    )
    _stub("pytorch_lightning.callbacks", ModelCheckpoint=object, TQDMProgressBar=object)
    _stub("pytorch_lightning.loggers", TensorBoardLogger=object)
    _stub("pytorch_lightning.plugins", DDPPlugin=object)
    _stub("torch_optimizer")
    _stub("cv2", COLORMAP_JET=2)
    # eval.py imports two drawing helpers; they are never called by batched_inference
    _stub("utils.visualization", visualize_depth=None, visualize_rgb_map_global=None)
    tv = _stub("torchvision")
    tv.transforms = _stub("torchvision.transforms")
    tv.utils = _stub("torchvision.utils")
    k = _stub("kornia", create_meshgrid=_create_meshgrid)
    k.losses = _stub("kornia.losses", ssim=None)
    _stub("imageio")
    ds = _stub("datasets", dataset_dict={})
    ds.depth_utils = _stub("datasets.depth_utils")
    for fn in ("save_pfm", "read_pfm"):
        setattr(ds.depth_utils, fn, None)


def get_hparams(**over):
    """opt.get_opts() with the reference defaults, then overridden."""
    install()
    argv = sys.argv
    sys.argv = ["x"]
    try:
        import opt

        hp = opt.get_opts()
    finally:
        sys.argv = argv
    for k_, v in over.items():
        setattr(hp, k_, v)
    return hp


def install_tcnn():
    """Stand-ins for the two CUDA-only encoders of models/mirror_nerf_tcnn.py, so that the file itself -- its MLPs (51-149)
    and `forward` (151-259): the [0,1] mapping, raw sigma = h[...,0], geo_feat = h[...,1:], the four heads, the three
    `detach` branches, the autograd normal -- imports and runs UNCHANGED in this container (fixtures G17):

    * `tinycudann.Encoding` (mirror_nerf_tcnn.py:39-49; un-vendored, un-pinned, CUDA-only): a torch module that evaluates a
      multiresolution hash grid from its `params` with the level geometry it is constructed with (n_levels,
      n_features_per_level, log2_hashmap_size, base_resolution, per_level_scale) -- the restatement in tests/torch_ref.py
      (`hashgrid_encode`: corner hashing / linear interpolation of models/gridencoder/src/gridencoder.cu:51-272, level
      sizes of models/gridencoder/grid.py:181-194), float32, twice differentiable.  The ENCODER'S interpolation is therefore
      NOT pinned by G17 (it is this repository's own reading); everything downstream of it is the reference's code.
    * `models.shencoder.SHEncoder` (models/encoding.py:71-74; the CUDA extension of models/shencoder/src/shencoder.cu): the
      degree-4 real spherical harmonics, pinned separately against scipy (tests/test_oracle_golden.py)."""
    install()
    import numpy as np
    import torch
    from torch import nn
    from tests import torch_ref as TR

    class Encoding(nn.Module):
        def __init__(self, n_input_dims, encoding_config, **_kw):
            super().__init__()
            c = dict(encoding_config)
            assert n_input_dims == 3 and c["otype"] == "HashGrid"
            L, F = int(c["n_levels"]), int(c["n_features_per_level"])
            offsets, off = [], 0
            for i in range(L):                                           # grid.py:181-194
                res = int(np.ceil(c["base_resolution"] * c["per_level_scale"] ** i))
                n = int(np.ceil(min(2 ** c["log2_hashmap_size"], (res + 1) ** 3) / 8) * 8)
                offsets.append(off)
                off += n
            offsets.append(off)
            self.encoding_config = c
            self.cfg = dict(offsets=np.array(offsets, dtype=np.int64), S=float(np.log2(c["per_level_scale"])),
                            H=int(c["base_resolution"]), n_levels=L, level_dim=F)
            self.n_output_dims = L * F
            self.params = nn.Parameter(torch.empty(off * F).uniform_(-1e-4, 1e-4))

        def forward(self, x):
            out = TR.hashgrid_encode(x, self.params.view(-1, self.cfg["level_dim"]), self.cfg)
            # `module.double()` runs (the generators' fp32-vs-fp64 noise floors): mirror_nerf_tcnn.py:225-227 calls `.float()`
            # on this output, which must then keep the float64 values
            return out.as_subclass(_KeepDouble) if out.dtype == torch.float64 else out

    class _KeepDouble(torch.Tensor):
        def float(self):
            return self.as_subclass(torch.Tensor)

    class SHEncoder(nn.Module):
        def __init__(self, input_dim=3, degree=4):
            super().__init__()
            assert input_dim == 3 and degree == 4
            self.input_dim, self.degree, self.output_dim = input_dim, degree, degree ** 2

        def forward(self, inputs, size=1):
            return TR.sh4(inputs / size)           # sphere_harmonics.py:83-84

    _stub("tinycudann", Encoding=Encoding)
    import models  # noqa: F401  (the reference package)
    _stub("models.shencoder", SHEncoder=SHEncoder)


def grid_encoder_class():
    """models/gridencoder/grid.py's `GridEncoder` with its compiled backend stubbed out: the constructor (level offsets,
    181-194) is plain Python and is what pins `hashgrid_config`'s table layout."""
    install()
    import models  # noqa: F401
    _stub("models.gridencoder.backend", _backend=None)
    from models.gridencoder.grid import GridEncoder
    return GridEncoder
