"""mirror_nerf_amd -- MI355X-native (gfx950) implementation of Mirror-NeRF's volumetric
rendering hot path behind the reference's own Python interfaces.

    from mirror_nerf_amd import render_rays, MirrorNeRF, Embedding      # models/rendering.py, models/mirror_nerf.py
    from mirror_nerf_amd import NeRFSystem, batched_inference          # train.py:102-348, eval.py:114-740
    from mirror_nerf_amd import get_loss                                # losses.py:258 (TotalLoss, fused value + gradient)

All arithmetic runs in libmnrf_hip.so (include/mnrf.h).  There is no CPU fallback.
"""
from .mirror_nerf import Embedding, MirrorNeRF, check_guard, reset_guard, set_precision, verify_split  # noqa: F401
from .mirror_nerf_tcnn import MirrorNeRFTcnn  # noqa: F401
from .rendering import render_rays, sample_pdf  # noqa: F401
from .recursion import NeRFSystem, batched_inference, render_rays_chunk_recursively  # noqa: F401
from .losses import TotalLoss, get_loss  # noqa: F401
from . import _lib  # noqa: F401

__all__ = ["Embedding", "MirrorNeRF", "render_rays", "sample_pdf", "NeRFSystem", "batched_inference",
           "render_rays_chunk_recursively", "TotalLoss", "get_loss"]
