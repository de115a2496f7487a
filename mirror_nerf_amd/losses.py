"""TotalLoss with the reference's constructor/forward contract (losses.py:201-259), evaluated by the HIP loss
kernels (csrc/mnrf_loss.hip): one call computes the five terms, their sum AND d(sum)/d(every input), so the autograd
node below only scales stored gradients.  No arithmetic in Python.

    crit = get_loss(hparams)                       # losses.py:258
    loss_sum, loss_dict = crit(results, batch, train_geometry_stage=False, epoch=epoch)   # train.py:439-446
    loss_sum.backward()

`loss_dict` holds the same keys as the reference's (detached 0-dim tensors, for logging); `loss_sum` carries the
graph.  PlaneConsistentLoss draws its point quadruples from the default CPU generator exactly like the reference
(4*floor(M/4) draws of torch.randint(high=M), "fine" before "coarse"; losses.py:96-107, 124-129) -- one batched
draw, the same sequence -- which costs one device->host read of M per step when that (optional) term is on.
Round 6: on the static training route and inside a captured step (batch["_plane_on_device"], or a capturing stream)
the term needs no host read: uniform numbers come from the DEVICE generator, sized for n_rays // 4 quadruples, and the
kernel forms every pick as floor(u * M) with M and the live quadruple count M // 4 from the count words of its own
call (MNRF_LOSS_PLANE_ON_DEVICE).  batch["_plane_u"] injects the numbers on either route (the host route then forms
the same picks from them), which is how tests hold the two routes to the same loss.
"""
import ctypes

import torch
from torch import nn

from . import _lib

_TYPS = ("coarse", "fine")
_F = ctypes.c_void_p


class _Args(ctypes.Structure):
    """MnrfLossArgs of include/mnrf.h."""
    _fields_ = [("rgb", _F * 2), ("mirror_mask", _F * 2), ("normal_dif", _F * 2), ("pred_normal", _F * 2),
                ("weights", _F * 2), ("x_surface", _F * 2), ("normal_fine", _F), ("n_samples", ctypes.c_int * 2),
                ("targets", _F), ("gt_mask", _F), ("rays", _F), ("valid_mask", _F), ("n_rays", ctypes.c_int64),
                ("plane_idx", _F * 2), ("plane_times", ctypes.c_int64 * 2), ("plane_u", _F * 2), ("plane_cap", ctypes.c_int64 * 2),
                ("w_color", ctypes.c_float), ("w_normal", ctypes.c_float), ("w_normal_reg", ctypes.c_float),
                ("w_mask", ctypes.c_float), ("w_plane", ctypes.c_float), ("flags", ctypes.c_uint),
                ("g_rgb", _F * 2), ("g_mirror_mask", _F * 2), ("g_normal_dif", _F * 2), ("g_pred_normal", _F * 2),
                ("g_weights", _F * 2), ("g_x_surface", _F * 2), ("g_normal_fine", _F), ("out", _F)]


FLAG_GEOMETRY_STAGE, FLAG_WO_MASK, FLAG_ONLY_INSIDE, FLAG_EXT_GRAD, FLAG_TCNN_BCE = 1, 2, 4, 8, 16
FLAG_USE_MASK, FLAG_USE_PLANE, FLAG_USE_NORMAL, FLAG_PLANE_ON_DEVICE = 32, 64, 128, 256
TERMS = ("color_loss", "mirror_mask_loss", "plane_consistent_loss", "normal_loss", "normal_reg_loss")

# (dict key pattern, struct field, gradient field); order fixes the order of the autograd inputs
_SLOTS = (("rgb_{}", "rgb", "g_rgb"), ("mirror_mask_{}", "mirror_mask", "g_mirror_mask"),
          ("normal_dif_{}", "normal_dif", "g_normal_dif"), ("pred_normal_{}", "pred_normal", "g_pred_normal"),
          ("weights_{}", "weights", "g_weights"), ("x_surface_{}", "x_surface", "g_x_surface"))


def _ptr(t):
    return None if t is None else t.data_ptr()


_UNIT = {}


def unit_gradient(device):
    """A constant 1.0 on `device` to hand to `loss.backward(gradient=...)`: torch then makes no tensor of ones for the root (one fill
    launch per step), and the fused loss recognises it by its address and returns its gradients unscaled (x 1.0 is the identity: one
    multi-tensor launch per step less).  Any other gradient -- a GradScaler's scale -- takes the general path."""
    key = str(device)
    t = _UNIT.get(key)
    if t is None:
        t = _UNIT[key] = torch.ones((), dtype=torch.float32, device=device)
    return t


class _TotalLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, batch, keys, *tensors):
        L = _lib.lib()
        inputs = dict(zip(keys, tensors))
        dev = batch["rgbs"].device
        a = _Args()
        keep = []

        def dev_f32(t):
            t = t.detach()
            if not t.is_cuda:
                raise RuntimeError("mirror_nerf_amd.losses runs on the GPU only (tensor is on %s)" % t.device)
            if t.dtype != torch.float32 or not t.is_contiguous():
                t = t.float().contiguous()
            keep.append(t)
            return t

        n = batch["rgbs"].reshape(-1, 3).shape[0]
        grads = {}
        for pat, field, gfield in _SLOTS:
            for ti, typ in enumerate(_TYPS):
                k = pat.format(typ)
                if k not in inputs:
                    continue
                if field == "mirror_mask":
                    # thresholded in place in one branch, as the reference does through .detach() (losses.py:27-33)
                    t = inputs[k].detach()
                    if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
                        raise RuntimeError(f"{k}: need a contiguous float32 CUDA tensor")
                    keep.append(t)
                else:
                    t = dev_f32(inputs[k])
                getattr(a, field)[ti] = _ptr(t)
                g = torch.empty_like(t)
                grads[k] = g
                getattr(a, gfield)[ti] = _ptr(g)
                if field in ("pred_normal", "weights"):
                    a.n_samples[ti] = t.shape[1]
        if "normal_fine" in inputs:
            t = dev_f32(inputs["normal_fine"])
            a.normal_fine = _ptr(t)
            a.n_samples[1] = t.shape[1]
            grads["normal_fine"] = torch.empty_like(t)
            a.g_normal_fine = _ptr(grads["normal_fine"])
        a.targets = _ptr(dev_f32(batch["rgbs"].reshape(-1, 3)))
        a.rays = _ptr(dev_f32(batch["rays"].reshape(-1, batch["rays"].shape[-1])[:, :8]))
        if batch["rays"].shape[-1] < 8:
            raise RuntimeError("batch['rays'] must hold (N, 8) rays")
        if "mirror_mask" in batch:
            a.gt_mask = _ptr(dev_f32(batch["mirror_mask"].reshape(-1)))
        if "valid_mask" in batch:
            vm = batch["valid_mask"].reshape(-1).to(torch.uint8).contiguous()
            keep.append(vm)
            a.valid_mask = _ptr(vm)
        a.n_rays = n
        a.w_color, a.w_normal, a.w_normal_reg = cfg["w_color"], cfg["w_normal"], cfg["w_normal_reg"]
        a.w_mask, a.w_plane = cfg["w_mask"], cfg["w_plane"]
        a.flags = cfg["flags"]
        buf = torch.empty(6, dtype=torch.float32, device=dev)
        out = buf[:6]             # (the six terms and the returned sum are two views of `buf`, not of each other: no clone launch)
        a.out = _ptr(out)

        times = [0, 0]
        s_max = max(a.n_samples[0], a.n_samples[1], 1)
        if (cfg["flags"] & FLAG_USE_PLANE) and "mirror_mask" in batch and any(f"x_surface_{t}" in inputs for t in _TYPS):
            plane_typs = [(ti, typ) for ti, typ in ((1, "fine"), (0, "coarse")) if f"x_surface_{typ}" in inputs]      # losses.py:124: fine first
            cap = n // 4
            u = batch.get("_plane_u")      # injected draws: (2, 4 * (n // 4)) uniform numbers, row 0 for "fine" (tests; both routes)
            if u is not None and (u.dtype != torch.float32 or tuple(u.shape) != (2, 4 * cap) or not u.is_cuda):
                raise ValueError("batch['_plane_u']: a float32 CUDA tensor of shape (2, 4 * (n_rays // 4))")
            on_device = bool(batch.get("_plane_on_device")) or torch.cuda.is_current_stream_capturing()
            if on_device and cap > 0:
                # no host read (static route, captured step): the kernel takes M = #GT-mirror rows from the count words of its own
                # call, draws row floor(u * M) for every pick of the M // 4 live quadruples and divides by M // 4; an invalid GT entry
                # switches the term off there (losses.py:116-119).  Draws: the DEVICE generator (the reference's are the CPU
                # generator's torch.randint: another sequence of the same distribution up to the 2^-24 grid of u)
                if u is None:
                    u = torch.rand(2, 4 * cap, device=dev)
                u = u.contiguous()
                keep.append(u)
                a.flags = cfg["flags"] | FLAG_PLANE_ON_DEVICE
                for k, (ti, _typ) in enumerate(plane_typs):
                    a.plane_u[ti] = u[k].data_ptr()
                    a.plane_cap[ti] = cap
                    times[ti] = cap
            elif not on_device:
                ws0 = torch.empty(L.mnrf_loss_workspace_floats(n, 1, 1, 0), dtype=torch.float32, device=dev)
                _lib.check(L.mnrf_loss_count(ctypes.byref(a), _lib.ptr(ws0), _lib.stream()), "mnrf_loss_count")
                n_invalid, m = (int(v) for v in ws0[:2].tolist())      # the one host read of this term
                if n_invalid == 0 and m // 4 > 0:
                    for k, (ti, typ) in enumerate(plane_typs):
                        if u is not None:      # the kernel's own expression on the injected numbers (fp32 product, truncation, clamp)
                            idx = (u[k, :4 * (m // 4)] * float(m)).to(torch.int64).clamp_(max=m - 1)
                        else:
                            idx = torch.randint(high=m, size=(4 * (m // 4),)).to(dev)
                        keep.append(idx)
                        a.plane_idx[ti] = _ptr(idx)
                        a.plane_times[ti] = times[ti] = m // 4
        ws = torch.empty(L.mnrf_loss_workspace_floats(n, a.n_samples[0] or 1, s_max, max(times)), dtype=torch.float32, device=dev)
        _lib.check(L.mnrf_total_loss(ctypes.byref(a), _lib.ptr(ws), _lib.stream()), "mnrf_total_loss")
        ctx.grads = [grads.get(k) for k in keys]
        ctx.shapes = [t.shape for t in tensors]
        ctx.mark_non_differentiable(out)
        ctx.set_materialize_grads(False)
        return buf[5], out

    @staticmethod
    def backward(ctx, g_sum, _g_out):
        # one multi-tensor launch for all inputs (four separate multiplies sat at the very start of the backward pass, where the
        # device has nothing else queued)
        if ctx.grads is None:
            # the unit-gradient path hands the STORED gradient tensors to autograd (no copy): a second pass over this node
            # (retain_graph=True) would alias buffers the first pass's consumers may have written in place
            raise RuntimeError("mirror_nerf_amd.losses: the fused loss was already back-propagated once (its stored gradients were "
                               "handed over); evaluate the loss again instead of retain_graph=True")
        if g_sum is None:
            return (None, None, None) + (None,) * len(ctx.grads)
        grads, ctx.grads = ctx.grads, None
        unit = _UNIT.get(str(g_sum.device))
        if unit is not None and g_sum.data_ptr() == unit.data_ptr():      # losses.unit_gradient: the factor is exactly 1
            return (None, None, None) + tuple(None if g is None else g.reshape(shape) for g, shape in zip(grads, ctx.shapes))
        have = [g for g in grads if g is not None]
        scaled = iter(torch._foreach_mul(have, g_sum) if have else [])
        outs = [None if g is None else next(scaled).reshape(shape) for g, shape in zip(grads, ctx.shapes)]
        return (None, None, None) + tuple(outs)


class TotalLoss(nn.Module):
    """losses.py:201-255."""

    def __init__(self, hparams):
        super().__init__()
        self.hparams = hparams

    def forward(self, inputs, batch, train_geometry_stage=False, epoch=-1):
        hp = self.hparams
        g = lambda name, default: getattr(hp, name, default)  # noqa: E731
        flags = FLAG_EXT_GRAD
        flags |= FLAG_GEOMETRY_STAGE if train_geometry_stage else 0
        flags |= FLAG_WO_MASK if g("woMaskRGBtoBlack", False) else 0
        flags |= FLAG_ONLY_INSIDE if g("normal_loss_only_inside_mirror", False) else 0
        flags |= FLAG_TCNN_BCE if g("model_type", "nerf") == "nerf_tcnn" else 0
        use = {"color_loss": True,
               "mirror_mask_loss": (not train_geometry_stage) or epoch >= g("train_mirror_mask_start_epoch", 2),
               "plane_consistent_loss": epoch >= g("smooth_mirror_start_epoch", 2) and g("use_plane_consistent_loss", False),
               "normal_loss": (not train_geometry_stage) or epoch >= g("train_normal_start_epoch", 1)}
        use["normal_reg_loss"] = use["normal_loss"]
        flags |= FLAG_USE_MASK if use["mirror_mask_loss"] else 0
        flags |= FLAG_USE_PLANE if use["plane_consistent_loss"] else 0
        flags |= FLAG_USE_NORMAL if use["normal_loss"] else 0
        cfg = dict(w_color=float(g("color_loss_weight", 1.0)), w_normal=float(g("normal_loss_weight", 1e-4)),
                   w_normal_reg=float(g("normal_reg_loss_weight", 0.1)), w_mask=float(g("mirror_mask_loss_weight", 0.1)),
                   w_plane=float(g("plane_consistent_loss_weight", 0.1)), flags=flags)
        keys = [pat.format(t) for pat, _, _ in _SLOTS for t in _TYPS if pat.format(t) in inputs]
        if "normal_fine" in inputs:
            keys.append("normal_fine")
        loss_sum, out = _TotalLossFn.apply(cfg, batch, keys, *[inputs[k] for k in keys])
        loss_dict = {name: out[i] for i, name in enumerate(TERMS) if use[name]}
        return loss_sum, loss_dict


def get_loss(hparams):
    """losses.py:258-259."""
    return TotalLoss(hparams)
