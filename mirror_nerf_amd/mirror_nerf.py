"""Embedding and MirrorNeRF with the reference's constructor/forward signatures
(models/mirror_nerf.py:6-38, 41-212), evaluated by the fused HIP field kernel.

The modules own ordinary `nn.Linear` parameters under the reference's names, built in the
reference's construction order (so a given `torch.manual_seed` yields the same weights and
reference checkpoints load with `load_state_dict`).  `forward` does no arithmetic in
Python: it hands pointers to `mnrf_field_forward`.  Outputs carry no autograd history
(inference / forward path; see DESIGN.md "out of scope this round").
"""
import os

import torch
from torch import nn

from . import _lib
from .weights import packed_of


class Embedding(nn.Module):
    """models/mirror_nerf.py:6-38.  x (B, f) -> (B, f*(2*N_freqs+1))."""

    def __init__(self, N_freqs, logscale=True):
        super().__init__()
        if not logscale:
            raise NotImplementedError("only logscale=True (the reference default) is implemented")
        self.N_freqs = N_freqs
        self.freq_bands = 2 ** torch.linspace(0, N_freqs - 1, N_freqs) if N_freqs > 0 else torch.zeros(0)

    def forward(self, x):
        x = x.float().contiguous()
        n, c = x.shape
        out = torch.empty(n, c * (2 * self.N_freqs + 1), dtype=torch.float32, device=x.device)
        if n:
            _lib.check(_lib.lib().mnrf_embed(_lib.ptr(x), n, c, self.N_freqs, _lib.ptr(out), _lib.stream()),
                       "mnrf_embed")
        return out


# Arithmetic of the field kernel's Linears:
#   "split" -- every fp32 operand carried as a hi/lo f16 pair on the f16 matrix pipe with fp32 accumulation
#              (MNRF_SPLIT_F16; agrees with the fp32 chain to ~3e-6, 3.2x faster; default; see
#              csrc/mnrf_field_split.inc);
#   "fp32"  -- v_mfma_f32_16x16x4_f32, bit-for-bit an fp32 fmaf chain.
# The training path (forward with saved activations, activation gradients, second-order pass) follows the same switch.
PRECISION = os.environ.get("MNRF_PRECISION", "split")


def set_precision(mode):
    """Select the arithmetic of the inference field kernel: "fp32" or "split"."""
    global PRECISION
    if mode not in ("fp32", "split", "split_h2", "split_h2x", "split_h1"):
        raise ValueError("precision must be 'fp32' or 'split'")
    PRECISION = mode


def verify_split(module, n=8192, bound=4.0, seed=0):
    """Evaluate `module` on n random positions of [-bound, bound]^3 with both arithmetics and return the largest
    difference of each output relative to max(1, |fp32 output|).  ~3e-6 is normal; a large value means some activation
    left the range the f16 hi/lo pairs carry (> 1.3e5) -- use set_precision("fp32") for that model."""
    global PRECISION
    dev = next(module.parameters()).device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    x = (torch.rand(n, 3, device=dev, generator=g) * 2 - 1) * bound
    d = torch.nn.functional.normalize(torch.randn(n, 3, device=dev, generator=g), dim=1)
    de = Embedding(4)(d)
    old, out = PRECISION, {}
    try:
        for mode in ("fp32", "split"):
            PRECISION = mode
            out[mode] = field_forward(module, n, xyz=x.contiguous(), xyz_stride=3, spr=1, dir_emb=de, dir_stride=27)
    finally:
        PRECISION = old
    return {k: float((out["fp32"][k] - out["split"][k]).abs().max() / max(1.0, float(out["fp32"][k].abs().max())))
            for k in out["fp32"]}


# When a list is installed here, every field-kernel launch is bracketed by two events on the
# launching stream and (flags, B, start, end) is appended -- bench.py reads kernel time from it.
LAUNCH_LOG = None

# algorithmic work per sample (SURVEY 8d / BASELINE.md section 3), FLOP = 2*MAC
FLOP_FULL = 1318912
FLOP_SIGMA = 982528
FLOP_GRAD = 982528      # density-gradient pass: the trunk GEMMs once more, transposed


def field_forward(module, B, *, xyz=None, xyz_stride=3, rays=None, z_vals=None, spr=1, dir_emb=None,
                  dir_stride=27, sigma_only=False, grad_normal=False, want_geo=False, device=None):
    """Run the fused field kernel; returns flat per-sample tensors (sigma (B,), rgb (B,3), ...)."""
    packed = packed_of(module)
    dev = packed.device
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
    out = {"sigma": f(B)}
    has_normal_head = True
    if has_normal_head and not sigma_only:
        out["pred_normal"] = f(B, 3)
    if not sigma_only:
        out["rgb"] = f(B, 3)
        out["is_mirror"] = f(B)
    if grad_normal:
        out["normal"] = f(B, 3)
    if want_geo:
        out["geo_feat"] = f(B, 256)
    flags = (_lib.MNRF_SIGMA_ONLY if sigma_only else 0) | (_lib.MNRF_GRAD_NORMAL if grad_normal else 0)
    if PRECISION == "split":
        flags |= _lib.MNRF_SPLIT_F16
    elif PRECISION == "split_h2":    # experiments only: force 16 KiB chunks
        flags |= _lib.MNRF_SPLIT_F16 | 8
    elif PRECISION == "split_h2x":   # experiments only: force 32 KiB chunks
        flags |= _lib.MNRF_SPLIT_F16 | 16
    elif PRECISION == "split_h1":    # experiments only (library built with -DMNRF_EXP_H1)
        flags |= _lib.MNRF_SPLIT_F16 | 24
    p = _lib.ptr
    if LAUNCH_LOG is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(_lib.lib().mnrf_field_forward(
        p(packed), flags, B, p(xyz), xyz_stride, p(rays), p(z_vals), spr, p(dir_emb), dir_stride,
        p(out["sigma"]), p(out.get("rgb")), p(out.get("pred_normal")), p(out.get("is_mirror")),
        p(out.get("normal")), p(out.get("geo_feat")), _lib.stream()), "mnrf_field_forward")
    if LAUNCH_LOG is not None:
        e1.record()
        LAUNCH_LOG.append((flags, B, e0, e1))
    return out


class MirrorNeRF(nn.Module):
    """models/mirror_nerf.py:41-212 (same constructor, parameter names and forward contract)."""

    def __init__(self, D=8, W=256, in_channels_xyz=63, in_channels_dir=27, skips=[4], **kwargs):
        super().__init__()
        self.D, self.W = D, W
        self.in_channels_xyz, self.in_channels_dir = in_channels_xyz, in_channels_dir
        self.skips = skips
        for i in range(D):
            if i == 0:
                layer = nn.Linear(in_channels_xyz, W)
            elif i in skips:
                layer = nn.Linear(W + in_channels_xyz, W)
            else:
                layer = nn.Linear(W, W)
            setattr(self, f"xyz_encoding_{i+1}", nn.Sequential(layer, nn.ReLU(True)))
        self.geo_feat_dim = W
        self.xyz_encoding_final = nn.Linear(W, W)
        self.dir_encoding = nn.Sequential(nn.Linear(W + in_channels_dir, W // 2), nn.ReLU(True))
        self.sigma = nn.Linear(W, 1)
        self.rgb = nn.Sequential(nn.Linear(W // 2, 3), nn.Sigmoid())
        self.predict_normal = kwargs.get("predict_normal", False)
        if self.predict_normal:
            self.hidden_dim_normal = W // 2
            self.normal_net = nn.Sequential(nn.Linear(W, W // 2), nn.Linear(W // 2, 3))
        self.predict_mirror_mask = kwargs.get("predict_mirror_mask", False)
        if self.predict_mirror_mask:
            self.hidden_dim_is_mirror = W // 2
            self.is_mirror_net = nn.Sequential(nn.Linear(W, W // 2), nn.LeakyReLU(inplace=True),
                                               nn.Linear(W // 2, 1), nn.Sigmoid())
        if (D, W, in_channels_xyz, in_channels_dir, list(skips)) != (8, 256, 63, 27, [4]) or not (
                self.predict_normal and self.predict_mirror_mask):
            raise NotImplementedError(
                "the HIP field kernel covers the reference default: D=8, W=256, 63/27 input channels, "
                "skips=[4], predict_normal=True, predict_mirror_mask=True")

    def forward(self, x, compute_normal=True, sigma_only=False, embedding_xyz=None, embedding_dir=None,
                mirror_mask=None, detach_density_outside_mirror_for_mask_loss=False,
                detach_density_for_mask_loss=False, detach_density_for_normal_loss=False):
        """x: (B,3) when sigma_only else (B, 3+27) = [raw xyz, embedded dir] (mirror_nerf.py:130-133).
        The three detach_* flags only alter gradients in the reference; values are identical."""
        if embedding_xyz is None or getattr(embedding_xyz, "N_freqs", None) != 10:
            raise NotImplementedError("embedding_xyz must be Embedding(10) (63 input channels)")
        x = x.float().contiguous()
        B = x.shape[0]
        if sigma_only:
            if x.shape[1] != 3:
                raise RuntimeError(f"sigma_only expects (B,3), got {tuple(x.shape)}")
        elif x.shape[1] != 3 + self.in_channels_dir:
            raise RuntimeError(f"expected (B,{3 + self.in_channels_dir}), got {tuple(x.shape)}")
        ld = x.shape[1]
        o = field_forward(self, B, xyz=x, xyz_stride=ld, spr=1,
                          dir_emb=None if sigma_only else x.view(-1)[3:], dir_stride=ld,
                          sigma_only=sigma_only, grad_normal=compute_normal, want_geo=True) if B else None
        out = {}
        dev = x.device
        z = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
        if compute_normal:
            out["normal"] = o["normal"] if B else z(0, 3)
        out["sigma"] = o["sigma"].view(B, 1) if B else z(0, 1)
        out["geo_feat"] = o["geo_feat"] if B else z(0, 256)
        if sigma_only:
            # the reference evaluates normal_net here as well (mirror_nerf.py:154-161)
            if B:
                xx = torch.cat([x, torch.zeros(B, self.in_channels_dir, device=dev)], 1).contiguous()
                o2 = field_forward(self, B, xyz=xx, xyz_stride=30, spr=1, dir_emb=xx.view(-1)[3:], dir_stride=30)
                out["pred_normal"] = o2["pred_normal"]
            else:
                out["pred_normal"] = z(0, 3)
        else:
            out["pred_normal"] = o["pred_normal"] if B else z(0, 3)
            out["rgb"] = o["rgb"] if B else z(0, 3)
            out["is_mirror"] = o["is_mirror"].view(B, 1) if B else z(0, 1)
        return out
