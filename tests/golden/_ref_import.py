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
