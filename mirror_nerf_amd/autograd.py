"""torch.autograd.Function wrappers: forward and backward are both HIP kernels of
libmnrf_hip.so; autograd only chains them (training path, train.py:437-439 `loss.backward()`)."""
import torch

from . import _lib


def _c(t):
    return None if t is None else t.contiguous()


class CompositeFn(torch.autograd.Function):
    """mnrf_composite / mnrf_composite_backward (models/rendering.py:181-264, 362-367).
    Differentiable inputs: rays (through x_surface), sigma, rgb, is_mirror, pred_normal, normal."""

    @staticmethod
    def forward(ctx, rays, sigma, z, noise, rgb, is_mirror, pred_normal, normal, white_back):
        N, S = z.shape
        dev = z.device
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
        rays, sigma, z = _c(rays), _c(sigma), _c(z)
        noise, rgb, is_mirror, pred_normal, normal = map(_c, (noise, rgb, is_mirror, pred_normal, normal))
        weights, opacity = f(N, S), f(N)
        full = rgb is not None
        rgb_map = f(N, 3) if full else None
        depth = f(N) if full else None
        mask = f(N) if (full and is_mirror is not None) else None
        sn = f(N, 3) if (full and pred_normal is not None) else None
        sng = f(N, 3) if (full and normal is not None) else None
        nd = f(N) if (sn is not None and sng is not None) else None
        xs = f(N, 3) if full else None
        p = _lib.ptr
        if N:
            _lib.check(_lib.lib().mnrf_composite(
                p(rays), N, S, p(sigma), p(z), p(noise), p(rgb), p(is_mirror), p(pred_normal), p(normal),
                int(bool(white_back)), p(weights), p(opacity), p(rgb_map), p(depth), p(mask), p(sn), p(sng), p(nd),
                p(xs), _lib.stream()), "mnrf_composite")
        ctx.save_for_backward(rays, sigma, z, noise, rgb, is_mirror, pred_normal, normal, depth)
        ctx.white_back = bool(white_back)
        ctx.present = (rgb_map is not None, depth is not None, mask is not None, sn is not None, sng is not None,
                       nd is not None, xs is not None)
        outs = (weights, opacity, rgb_map, depth, mask, sn, sng, nd, xs)
        ctx.mark_non_differentiable(*[o for o in () if o is not None])
        return outs

    @staticmethod
    def backward(ctx, g_w, g_op, g_rgb, g_depth, g_mask, g_sn, g_sng, g_nd, g_xs):
        rays, sigma, z, noise, rgb, is_mirror, pred_normal, normal, depth = ctx.saved_tensors
        N, S = z.shape
        dev = z.device
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
        need = ctx.needs_input_grad
        d_rays = f(N, 8) if (need[0] and depth is not None) else None
        d_sigma = f(N, S) if need[1] else None
        d_rgb = f(N * S, 3) if (need[4] and rgb is not None) else None
        d_m = f(N * S) if (need[5] and is_mirror is not None) else None
        d_pn = f(N * S, 3) if (need[6] and pred_normal is not None) else None
        d_n = f(N * S, 3) if (need[7] and normal is not None) else None
        p = _lib.ptr
        g = [None if t is None else t.contiguous().float() for t in (g_w, g_op, g_rgb, g_depth, g_mask, g_sn, g_sng, g_nd, g_xs)]
        if N:
            _lib.check(_lib.lib().mnrf_composite_backward(
                p(rays), N, S, p(sigma), p(z), p(noise), p(rgb), p(is_mirror), p(pred_normal), p(normal),
                int(ctx.white_back), None, p(depth), p(g[0]), p(g[1]), p(g[2]), p(g[3]), p(g[4]), p(g[5]), p(g[6]),
                p(g[7]), p(g[8]), p(d_sigma), p(d_rgb), p(d_m), p(d_pn), p(d_n), p(d_rays), _lib.stream()),
                "mnrf_composite_backward")

        def like(d, ref):
            return None if d is None else d.view(ref.shape)
        return (d_rays, like(d_sigma, sigma), None, None, like(d_rgb, rgb) if rgb is not None else None,
                like(d_m, is_mirror) if is_mirror is not None else None,
                like(d_pn, pred_normal) if pred_normal is not None else None,
                like(d_n, normal) if normal is not None else None, None)
