"""torch.autograd.Function wrappers: forward and backward are both HIP kernels of
libmnrf_hip.so; autograd only chains them (training path, train.py:437-439 `loss.backward()`)."""
import torch

from . import _lib


import os

# MNRF_FOLD_GRADS=0: leave the accumulation of a module's parameter gradients over its evaluations to autograd
FOLD_GRADS = os.environ.get("MNRF_FOLD_GRADS", "1") != "0"
# MNRF_DW_PLANES=0: the round-1/2 weight-gradient route of the split arithmetic (fp32 rows of saved activations and of dY, one
# set of bf16 x 6 GEMM launches per evaluation) instead of operand planes + ONE GEMM launch per module and pass (mnrf_dwp.h)
RAY_GRADS_KERNEL = os.environ.get("MNRF_RAY_GRADS", "1") != "0"      # 0: the torch ops it replaced (A/B measurements)
DW_PLANES = os.environ.get("MNRF_DW_PLANES", "1") != "0"
# MNRF_DW_PLANES_HALF=1 (round 6, opt-in, NOT exact): the activation gradients travel to the weight-gradient GEMM as one f16 per
# element instead of a hi/lo pair (include/mnrf.h MNRF_PLANES_Y_HALF): half the backward kernel's plane traffic, 3/4 of the GEMM's.
# Emulated on the reference in float64 first (scripts/exp_half_planes.py): 1.2e-4 .. 7.4e-4 of a tensor's largest entry, inside the
# 1e-3 bar of the gradient fixtures with a margin of 1.3 x at worst -- which is why it is not the default.
DW_PLANES_HALF = os.environ.get("MNRF_DW_PLANES_HALF", "0") == "1"
DW2_PLANES = os.environ.get("MNRF_DW2_PLANES", "1") != "0"    # 0: the second-order term on fp32 rows (mnrf_field_backward2)


def _c(t):
    return None if t is None else t.contiguous()


class CompositeFn(torch.autograd.Function):
    """mnrf_composite / mnrf_composite_backward (models/rendering.py:181-264, 362-367).
    Differentiable inputs: rays (through x_surface), sigma, rgb, is_mirror, pred_normal, normal."""

    @staticmethod
    def forward(ctx, rays, sigma, z, noise, rgb, is_mirror, pred_normal, normal, white_back, detach=0, keep_mirror=None, n_live=None,
                resample=None):
        """detach: MNRF_DETACH_W_MASK | MNRF_DETACH_W_NORMAL; keep_mirror: (N,) float, 0 = that ray's mirror mask sees
        weights.detach() (models/rendering.py:223-247: the --detach_density_* options; values are unaffected).
        n_live: device int32 (1,) or None -- the rays that exist, N being the capacity (include/mnrf.h "live row counts").
        resample: None, or (u, n_importance) -- the resampling that follows a coarse pass in the same launch (mnrf_composite_sample_n):
        a tenth output, z_fine (N, S + n_importance), not differentiable (the reference detaches the weights there, rendering.py:335)."""
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
        z_fine = None
        if N and resample is not None:
            u, n_imp = resample
            u = u.float().contiguous()
            z_fine = f(N, S + n_imp)
            _lib.check(_lib.lib().mnrf_composite_sample_n(
                p(rays), N, S, p(sigma), p(z), p(noise), p(rgb), p(is_mirror), p(pred_normal), p(normal),
                int(bool(white_back)), p(weights), p(opacity), p(rgb_map), p(depth), p(mask), p(sn), p(sng), p(nd),
                p(xs), p(u), 1 if u.dim() == 2 else 0, int(n_imp), p(z_fine), p(n_live), _lib.stream()), "mnrf_composite_sample_n")
        elif N:
            _lib.check(_lib.lib().mnrf_composite_n(
                p(rays), N, S, p(sigma), p(z), p(noise), p(rgb), p(is_mirror), p(pred_normal), p(normal),
                int(bool(white_back)), p(weights), p(opacity), p(rgb_map), p(depth), p(mask), p(sn), p(sng), p(nd),
                p(xs), p(n_live), _lib.stream()), "mnrf_composite")
        ctx.n_live = n_live
        ctx.save_for_backward(rays, sigma, z, noise, rgb, is_mirror, pred_normal, normal, depth)
        ctx.white_back = bool(white_back)
        ctx.detach, ctx.keep_mirror = int(detach), _c(keep_mirror)
        ctx.set_materialize_grads(False)
        ctx.present = (rgb_map is not None, depth is not None, mask is not None, sn is not None, sng is not None,
                       nd is not None, xs is not None)
        outs = (weights, opacity, rgb_map, depth, mask, sn, sng, nd, xs)
        if resample is not None:
            if z_fine is None:
                z_fine = f(N, S + resample[1])
            ctx.mark_non_differentiable(z_fine)
            return outs + (z_fine,)
        return outs

    @staticmethod
    def backward(ctx, g_w, g_op, g_rgb, g_depth, g_mask, g_sn, g_sng, g_nd, g_xs, _g_zf=None):
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
        # dL/dnormal is identically zero unless a loss reads surface_normal_grad / normal_dif: returning None
        # then lets FieldFn skip the second-order pass
        d_n = f(N * S, 3) if (need[7] and normal is not None and (g_sng is not None or g_nd is not None)) else None
        p = _lib.ptr
        g = [None if t is None else t.contiguous().float() for t in (g_w, g_op, g_rgb, g_depth, g_mask, g_sn, g_sng, g_nd, g_xs)]
        if N:
            _lib.check(_lib.lib().mnrf_composite_backward_n(
                p(rays), N, S, p(sigma), p(z), p(noise), p(rgb), p(is_mirror), p(pred_normal), p(normal),
                int(ctx.white_back), None, p(depth), p(g[0]), p(g[1]), p(g[2]), p(g[3]), p(g[4]), p(g[5]), p(g[6]),
                p(g[7]), p(g[8]), p(d_sigma), p(d_rgb), p(d_m), p(d_pn), p(d_n), p(d_rays), ctx.detach, p(ctx.keep_mirror),
                p(ctx.n_live), _lib.stream()), "mnrf_composite_backward")

        def like(d, ref):
            return None if d is None else d.view(ref.shape)
        return (d_rays, like(d_sigma, sigma), None, None, like(d_rgb, rgb) if rgb is not None else None,
                like(d_m, is_mirror) if is_mirror is not None else None,
                like(d_pn, pred_normal) if pred_normal is not None else None,
                like(d_n, normal) if normal is not None else None, None, None, None, None, None)


class FieldFn(torch.autograd.Function):
    """mnrf_field_forward_train / mnrf_field_backward: the fused MirrorNeRF evaluation with a
    hand-written backward (activation gradients by transposed MFMA chains, weight gradients by
    split-K MFMA GEMMs over the saved activations).

    apply(module, spr, xyz, rays, z_vals, dir_emb, want_normal, *params) ->
        sigma (B), rgb (B,3), pred_normal (B,3), is_mirror (B), normal (B,3 or empty)
    `want_normal` may also be a tuple (want_normal, cut_flags, keep_mirror[, n_live]): cut_flags = MNRF_CUT_NORMAL_HEAD |
    MNRF_CUT_MIRROR_HEAD and keep_mirror (per ray / per row of xyz; 0 = cut the mirror head there) make those heads see
    geo_feat.detach() (models/mirror_nerf.py:154-183, the --detach_density_* options); n_live: device int32 (1,) = the rays
    (rows) that exist, the tensors being sized for a capacity (include/mnrf.h "live row counts"; planes route only).
    Positions come from `xyz` (B,>=3 columns, row stride = its row length) or from rays (N,8) and
    z_vals (N,spr).  `dir_emb`: (B/spr, 27) view encoding.  `params`: the module's 32 parameters in
    state_dict order (so that autograd routes their gradients).
    The gradient arriving at `normal` (the normalised density gradient) is a second-order term:
    it is propagated by mnrf_field_backward2 (tangent pass + weight-gradient GEMMs)."""

    @staticmethod
    def forward(ctx, module, spr, xyz, rays, z_vals, dir_emb, want_normal, *params):
        from .weights import packed_of
        from . import mirror_nerf as _mn
        L = _lib.lib()
        ctx.cut, ctx.keep_mirror, ctx.n_live = 0, None, None
        if isinstance(want_normal, tuple):
            ctx.n_live = want_normal[3] if len(want_normal) > 3 else None
            want_normal, ctx.cut, ctx.keep_mirror = want_normal[0], int(want_normal[1]), _c(want_normal[2])
        packed = packed_of(module)
        dev = packed.device
        B = xyz.shape[0] if xyz is not None else rays.shape[0] * spr
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
        xyz, rays, z_vals, dir_emb = _c(xyz), _c(rays), _c(z_vals), _c(dir_emb)
        sigma, rgb, pn, mir = f(B), f(B, 3), f(B, 3), f(B)
        normal = f(B, 3) if want_normal else None
        split = _mn.precision_of(module).startswith("split")
        planes = split and DW_PLANES
        if ctx.n_live is not None and not planes:
            raise RuntimeError("a live row count (static training step) needs the split arithmetic with operand planes")
        if planes:     # the inputs of every Linear as hi/lo f16 operand tiles of the weight-gradient GEMM (mnrf_dwp.h)
            save_x = torch.empty(max(16, L.mnrf_train_planes_bytes(B)), dtype=torch.uint8, device=dev)
        else:
            save_x = f(max(1, L.mnrf_train_save_floats(B)))
        save_mask = torch.empty(max(1, L.mnrf_train_mask_words(B)), dtype=torch.int64, device=dev)
        save_inv = f(max(1, B))
        save_invj = f(max(1, B)) if want_normal else None
        p = _lib.ptr
        xs = xyz.shape[1] if xyz is not None else 3
        if B:
            _lib.check(L.mnrf_field_forward_train_n(
                p(packed), B, p(xyz), xs, p(rays), p(z_vals), spr, p(dir_emb), dir_emb.shape[1], p(sigma), p(rgb),
                p(pn), p(mir), p(normal), p(save_x), p(save_mask), p(save_inv), p(save_invj),
                (_lib.MNRF_SPLIT_F16 if split else 0) | (_lib.MNRF_TRAIN_PLANES if planes else 0), p(ctx.n_live), _lib.stream()),
                "mnrf_field_forward_train")
        ctx.split, ctx.planes = split, planes    # the backward follows the arithmetic (and the save format) of ITS forward
        ctx.packed = packed                      # ... and reads ITS weights: the image object (31 us of host time per look-up)
        ctx.module, ctx.spr, ctx.B = module, spr, B
        ctx.set_materialize_grads(False)   # an unused `normal` must arrive as None, not as zeros: it gates the second-order pass
        ctx.save_for_backward(xyz, rays, z_vals, rgb, pn, mir, save_x, save_mask, save_inv, normal, save_invj)
        # names / shapes of `params` (same order: weights.params_of), cached on the module next to the list they derive from
        from .weights import param_refs
        refs = param_refs(module)
        meta = module.__dict__.get("_mnrf_param_meta")
        if meta is None or meta[0] is not refs or len(meta[1]) != len(params):
            meta = (refs, [full for _, _, full in refs], [tuple(t.shape) for t in params])
            module.__dict__["_mnrf_param_meta"] = meta
        ctx.param_names, ctx.param_shapes = meta[1], meta[2]
        # how many evaluations of this module await their backward (primary + reflected rays: see backward).  A gradient
        # buffer still pending at FORWARD time belongs to a backward pass that died (an exception skips the engine's
        # end-of-pass callbacks): drop it together with its count
        if module.__dict__.pop("_mnrf_pending", None) is not None:
            module.__dict__["_mnrf_uses"] = 0
        module.__dict__["_mnrf_uses"] = module.__dict__.get("_mnrf_uses", 0) + 1
        if normal is None:
            normal = f(0, 3)
            ctx.mark_non_differentiable(normal)
        return sigma, rgb, pn, mir, normal

    @staticmethod
    def backward(ctx, g_sigma, g_rgb, g_pn, g_m, g_normal):
        import ctypes
        L = _lib.lib()
        xyz, rays, z_vals, rgb, pn, mir, save_x, save_mask, save_inv, normal, save_invj = ctx.saved_tensors
        B, spr = ctx.B, ctx.spr
        packed = ctx.packed
        dev = packed.device
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
        z = lambda t, *s: torch.zeros(*s, dtype=torch.float32, device=dev) if t is None else t.contiguous().float()  # noqa: E731
        if ctx.planes:      # a missing upstream gradient stays a null pointer there (no tensor of zeros, no memset)
            g_sigma, g_rgb, g_pn, g_m = [None if t is None else t.contiguous().float() for t in (g_sigma, g_rgb, g_pn, g_m)]
        else:
            g_sigma, g_rgb, g_pn, g_m = z(g_sigma, B), z(g_rgb, B, 3), z(g_pn, B, 3), z(g_m, B)
        # A module evaluated several times in one step (primary rays, then reflected rays: train.py:253-259) gets one
        # gradient per evaluation and autograd would add them with one kernel per parameter (64 launches per step).
        # Instead the evaluations of a pass work on ONE private set of gradient tensors (views of one flat buffer) and
        # return None; the LAST pending evaluation (count from the forwards) hands the complete tensors to autograd, which
        # sums them with whatever other consumers of the parameters contribute (a weight regulariser ...).  Nothing is
        # handed over before it is complete, so autograd never holds a tensor that is still being added into.  An
        # end-of-pass callback, registered by the first evaluation of every pass, covers a mis-count (an evaluation whose
        # outputs never reached the loss): gradients still pending then are added to `.grad` directly.
        #   planes route (split arithmetic, default): an evaluation only runs its activation-gradient kernel and leaves
        #   its operand planes on the module's TAPE; the last one launches the weight-gradient GEMM once over the whole tape;
        #   rows route (fp32 arithmetic, MNRF_DW_PLANES=0): every evaluation runs its own GEMMs, adding into the tensors.
        mod = ctx.module
        uses = max(0, mod.__dict__.get("_mnrf_uses", 1) - 1)
        mod.__dict__["_mnrf_uses"] = uses
        fold = FOLD_GRADS or ctx.planes
        st = mod.__dict__.get("_mnrf_pending") if fold else None
        standalone = False
        if st is not None and st.shapes != ctx.param_shapes:
            st = None                        # (tensors of another parameter set pending: this evaluation stands alone and
            fold = False                     #  leaves that set -- its partial sums and its tape -- where it is)
            standalone = True
        first = st is None
        if first:
            st = _Pending(ctx.param_names, ctx.param_shapes, dev)
        arr = st.pointers()
        need = ctx.needs_input_grad
        want_xyz = (xyz is not None and need[2]) or (rays is not None and need[3])
        d_xyz = f(B, 3) if want_xyz else None
        d_dir = f(B, 32) if need[5] else None
        p = _lib.ptr
        xs = xyz.shape[1] if xyz is not None else 3
        if B and ctx.planes:
            red = int(mod.__dict__.get("_mnrf_seed_reduction", 0))      # mirror_nerf._lower_gradient_scale
            dy = torch.empty(max(16, L.mnrf_train_dy_planes_bytes(B)), dtype=torch.uint8, device=dev)
            seed = torch.empty(1, dtype=torch.int32, device=dev)
            _lib.check(L.mnrf_field_backward_planes_n(
                p(packed), B, p(xyz), xs, p(rays), p(z_vals), spr, p(g_sigma), p(g_rgb), p(g_pn), p(g_m), p(rgb), p(pn),
                p(mir), p(save_mask), p(save_inv), p(dy), p(seed), p(d_xyz), p(d_dir), p(ctx.keep_mirror),
                ctx.cut | (red << 16) | (_lib.MNRF_PLANES_Y_HALF if DW_PLANES_HALF else 0),
                p(ctx.n_live), _lib.stream()), "mnrf_field_backward_planes")
            # (kind 0 | the gradient-scale reduction of this launch | dY as one f16)
            st.tape.append((save_x, dy, B, seed, (red << 8) | (0x1000 if DW_PLANES_HALF else 0), ctx.n_live, spr))
        elif B:
            ws = f(max(1, L.mnrf_train_workspace_floats(B)))
            _lib.check(L.mnrf_field_backward(
                p(packed), B, p(xyz), xs, p(rays), p(z_vals), spr, p(g_sigma), p(g_rgb), p(g_pn), p(g_m), p(rgb), p(pn),
                p(mir), p(save_x), p(save_mask), p(save_inv), p(ws), arr, p(d_xyz), p(d_dir), p(ctx.keep_mirror),
                (_lib.MNRF_SPLIT_F16 if ctx.split else 0) | ctx.cut | (_lib.MNRF_DW_ACCUMULATE if st.dirty else 0), _lib.stream()),
                "mnrf_field_backward")
            st.dirty = True
        if B and g_normal is not None and normal is not None and ctx.planes and DW2_PLANES:
            # second-order term through the density-gradient normal, planes route (round 4): the tangent pass leaves its
            # operands on the tape as one more entry (kind 1); the module's ONE weight-gradient GEMM contracts them too
            x2 = torch.empty(max(16, L.mnrf_train_planes2_bytes(B)), dtype=torch.uint8, device=dev)
            y2 = torch.empty(max(16, L.mnrf_train_dy_planes2_bytes(B)), dtype=torch.uint8, device=dev)
            jmax = torch.empty(1, dtype=torch.int32, device=dev)
            _lib.check(L.mnrf_field_backward2_planes_n(
                p(packed), B, p(xyz), xs, p(rays), p(z_vals), spr, p(g_normal.contiguous().float()), p(normal),
                p(save_invj), p(save_mask), p(x2), p(y2), p(jmax), p(d_xyz), p(ctx.n_live), _lib.stream()), "mnrf_field_backward2_planes")
            st.tape.append((x2, y2, B, jmax, 1, ctx.n_live, spr))
        elif B and g_normal is not None and normal is not None:   # rows route (fp32 arithmetic, MNRF_DW2_PLANES=0)
            if not st.dirty:         # it ADDS to the gradients (trunk weights, sigma.weight): they start from zero then
                st.flat.zero_()
                st.dirty = True
            ws2 = f(max(1, L.mnrf_train_workspace2_floats(B)))
            _lib.check(L.mnrf_field_backward2(
                p(packed), B, p(xyz), xs, p(rays), p(z_vals), spr, p(g_normal.contiguous().float()), p(normal),
                p(save_invj), p(save_mask), p(ws2), arr, p(d_xyz),
                _lib.MNRF_SPLIT_F16 if ctx.split else 0, _lib.stream()), "mnrf_field_backward2")
        hand_over = True
        if fold and uses > 0:
            hand_over = False                 # more evaluations of this module to come in this pass
            if first:
                mod.__dict__["_mnrf_pending"] = st

                def _end_of_pass(m=mod):
                    left = m.__dict__.pop("_mnrf_pending", None)
                    m.__dict__["_mnrf_uses"] = 0
                    if left is not None:      # never handed over (mis-count): deliver the sum ourselves
                        by = dict(zip(left.names, left.finish()))
                        from .weights import param_refs
                        for sub, pname, full in param_refs(m):
                            q, gq = sub._parameters[pname], by.get(full)
                            if gq is not None and q is not None and q.requires_grad:
                                q.grad = gq if q.grad is None else q.grad.add_(gq)
                torch.autograd.Variable._execution_engine.queue_callback(_end_of_pass)
        d_params = None
        if hand_over:
            if not standalone:
                mod.__dict__.pop("_mnrf_pending", None)
            d_params = st.finish()            # the weight-gradient GEMM over the tape (planes route); complete now
            mod.__dict__["_mnrf_flat_grad"] = st.flat     # dist.allreduce_gradients reduces this buffer in place when .grad aliases it
            st = None
        g_xyz = g_rays = g_de = None
        if d_xyz is not None:
            if xyz is not None:
                g_xyz = torch.zeros_like(xyz)
                g_xyz[:, :3] = d_xyz
            elif rays.shape[1] == 8 and RAY_GRADS_KERNEL:   # x = o + d*z  (rendering.py:302): dL/do = sum_s dL/dx, dL/dd = sum_s z dL/dx -- one kernel,
                g_rays = torch.empty_like(rays)      # together with the per-ray sum of the view-encoding gradient
                if d_dir is not None:
                    g_de = f(rays.shape[0], 27)
                _lib.check(L.mnrf_ray_grads_n(p(d_xyz), p(z_vals), p(d_dir), rays.shape[0], spr, p(g_rays), p(g_de), p(ctx.n_live),
                                              _lib.stream()), "mnrf_ray_grads")
            else:
                N = rays.shape[0]
                dx = d_xyz.view(N, spr, 3)
                g_rays = torch.zeros_like(rays)
                g_rays[:, 0:3] = dx.sum(1)
                g_rays[:, 3:6] = (dx * z_vals.view(N, spr, 1)).sum(1)
        if d_dir is not None and g_de is None:
            if rays is not None and xyz is None and B and RAY_GRADS_KERNEL:
                g_de = f(rays.shape[0], 27)
                _lib.check(L.mnrf_ray_grads_n(None, None, p(d_dir), rays.shape[0], spr, None, p(g_de), p(ctx.n_live), _lib.stream()),
                           "mnrf_ray_grads")
            else:
                g_de = d_dir.view(-1, spr, 32)[:, :, :27].sum(1)
        n_par = len(ctx.param_shapes)
        return (None, None, g_xyz, g_rays, None, g_de, None, *(d_params if hand_over else [None] * n_par))


class _Pending:
    """Gradient tensors of one module for one backward pass: views of ONE flat buffer in state_dict order of the 32 field
    parameters (an absent optional head keeps its slots as scratch), plus the tape of the planes route.  The kernels only need
    the 32 addresses (base + offset); the views are made when the gradients are handed over (creating them up front cost 70 us
    of host time in front of the first backward launch of every module, where the device has nothing else queued)."""
    _layout = None      # (names, sizes, byte offsets, total) of weights.PARAM_NAMES, computed once

    def __init__(self, names, shapes, dev):
        if _Pending._layout is None:
            from .weights import PARAM_NAMES, PARAM_SHAPES
            sizes = [int(torch.Size(PARAM_SHAPES[n]).numel()) for n in PARAM_NAMES]
            offs, off = [], 0
            for k in sizes:
                offs.append(off)
                off += k
            _Pending._layout = (list(PARAM_NAMES), sizes, offs, off)
        self.names, self.shapes = list(names), list(shapes)
        self.flat = torch.empty(_Pending._layout[3], dtype=torch.float32, device=dev)
        self._views = None
        self.tape = []          # (x_planes, dy_planes, B, seedmax, kind) per evaluation and order
        self.dirty = False      # the tensors hold a partial sum already (rows route, second-order pass)

    def pointers(self):
        """ctypes array of the 32 gradient addresses (what the C ABI's d_params takes)."""
        import ctypes
        base = self.flat.data_ptr()
        return (ctypes.c_void_p * _lib.N_PARAMS)(*[base + 4 * o for o in _Pending._layout[2]])

    def views(self):
        if self._views is None:
            from .weights import PARAM_SHAPES
            names, sizes, offs, _ = _Pending._layout
            self._views = {n: self.flat[o:o + k].view(PARAM_SHAPES[n]) for n, k, o in zip(names, sizes, offs)}
        return self._views

    def all32(self):
        v = self.views()
        return [v[n] for n in _Pending._layout[0]]

    def finish(self):
        """Complete the gradients (launch the weight-gradient GEMM over the tape) and return them in the module's order."""
        import ctypes
        L = _lib.lib()
        p = _lib.ptr
        tape, self.tape = self.tape, []
        if tape:
            arr = self.pointers()
            for g0 in range(0, len(tape), 8):          # mnrf_dw_planes takes up to 8 evaluations per call
                grp = tape[g0:g0 + 8]
                n = len(grp)
                xs = (ctypes.c_void_p * n)(*[t[0].data_ptr() for t in grp])
                ys = (ctypes.c_void_p * n)(*[t[1].data_ptr() for t in grp])
                bs = (ctypes.c_int64 * n)(*[t[2] for t in grp])
                sm = (ctypes.c_void_p * n)(*[t[3].data_ptr() for t in grp])
                kd = (ctypes.c_int * n)(*[t[4] for t in grp])      # 0: first-order planes, 1: second-order planes
                if any(len(t) > 5 and t[5] is not None for t in grp):
                    # an evaluation whose sample count lives on the device (the reflected rays of a static step): the GEMM's work
                    # plan is made on the device too (mnrf_dw_planes2_n), B = capacities
                    nl = (ctypes.c_void_p * n)(*[(t[5].data_ptr() if (len(t) > 5 and t[5] is not None) else None) for t in grp])
                    sp = (ctypes.c_int * n)(*[(t[6] if len(t) > 6 else 1) for t in grp])
                    ws = torch.empty(max(1, L.mnrf_dw_planes2_n_workspace_floats(n)), dtype=torch.float32, device=self.flat.device)
                    _lib.check(L.mnrf_dw_planes2_n(n, xs, ys, bs, nl, sp, sm, kd, p(ws), arr, 1 if self.dirty else 0, _lib.stream()),
                               "mnrf_dw_planes2_n")
                else:
                    ws = torch.empty(max(1, L.mnrf_dw_planes2_workspace_floats(n, bs, kd)), dtype=torch.float32, device=self.flat.device)
                    _lib.check(L.mnrf_dw_planes2(n, xs, ys, bs, sm, kd, p(ws), arr, 1 if self.dirty else 0, _lib.stream()), "mnrf_dw_planes2")
                self.dirty = True
        elif not self.dirty:
            self.flat.zero_()      # no samples at all in this pass
        from .weights import decanonical      # (a model with fewer encoding bands: its own column count, weights.canonical)
        views = self.views()
        return [decanonical(n, views[n], s) if n in views else torch.zeros(s, device=self.flat.device)
                for n, s in zip(self.names, self.shapes)]


class EmbedFn(torch.autograd.Function):
    """mnrf_embed / mnrf_embed_backward (Embedding.forward, models/mirror_nerf.py:20-38): the view encoding of
    reflected rays carries gradient back to the surface normal (train.py:205 "not detach()")."""

    @staticmethod
    def forward(ctx, x, n_freqs, n_live=None):
        x = x.float().contiguous()
        n, c = x.shape
        out = torch.empty(n, c * (2 * n_freqs + 1), dtype=torch.float32, device=x.device)
        if n:
            _lib.check(_lib.lib().mnrf_embed_n(_lib.ptr(x), n, c, n_freqs, _lib.ptr(out), _lib.ptr(n_live), _lib.stream()), "mnrf_embed")
        ctx.save_for_backward(x)
        ctx.n_freqs, ctx.n_live = n_freqs, n_live
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        n, c = x.shape
        gx = torch.empty_like(x)
        if n:
            _lib.check(_lib.lib().mnrf_embed_backward_n(_lib.ptr(x), _lib.ptr(g.contiguous().float()), n, c, ctx.n_freqs,
                                                        _lib.ptr(gx), _lib.ptr(ctx.n_live), _lib.stream()), "mnrf_embed_backward")
        return gx, None, None


class RayFanFn(torch.autograd.Function):
    """The head of a training render_rays call (models/rendering.py:275-300) and the tail of its backward, one launch each:
    apply(rays, n_freqs_dir, z_steps, N_samples, use_disp, perturb, prand, n_live) ->
        (rays_a, rays_b, rays_c, rays_d, dir_emb_a, dir_emb_b, z_vals)
    rays_* are `rays` itself (views: one per consumer -- two field evaluations, two compositing passes), dir_emb_a / _b ONE view
    encoding of the directions (one per field evaluation), z_vals the coarse depths (mnrf_ray_prologue_n).  The backward takes the
    gradient every consumer sends and returns their sum in one launch (mnrf_ray_fan_backward_n) where autograd would add the pieces
    pairwise, run the encoding's backward, pad it to eight columns and add again (eight launches per level of reflected rays,
    train.py:205 "not detach() to jointly optimize")."""

    @staticmethod
    def forward(ctx, rays, n_freqs, z_steps, n_samples, use_disp, perturb, prand, n_live=None):
        N = rays.shape[0]
        dev = rays.device
        dir_emb = torch.empty(N, 3 * (2 * n_freqs + 1), dtype=torch.float32, device=dev)
        z_vals = torch.empty(N, n_samples, dtype=torch.float32, device=dev)
        p = _lib.ptr
        _lib.check(_lib.lib().mnrf_ray_prologue_n(p(rays), N, n_freqs, p(z_steps), n_samples, int(bool(use_disp)), float(perturb), p(prand),
                                                  p(dir_emb), p(z_vals), p(n_live), _lib.stream()), "mnrf_ray_prologue_n")
        ctx.save_for_backward(rays)
        ctx.n_freqs, ctx.n_live = n_freqs, n_live
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(z_vals)
        return (rays.view_as(rays), rays.view_as(rays), rays.view_as(rays), rays.view_as(rays), dir_emb, dir_emb.view_as(dir_emb), z_vals)

    @staticmethod
    def backward(ctx, g0, g1, g2, g3, ga, gb, _gz):
        (rays,) = ctx.saved_tensors
        g = [None if t is None else t.contiguous().float() for t in (g0, g1, g2, g3, ga, gb)]
        if all(t is None for t in g):
            return (None,) * 8
        out = torch.empty_like(rays)
        p = _lib.ptr
        _lib.check(_lib.lib().mnrf_ray_fan_backward_n(p(g[0]), p(g[1]), p(g[2]), p(g[3]), p(rays), p(g[4]), p(g[5]), rays.shape[0],
                                                      ctx.n_freqs, p(out), p(ctx.n_live), _lib.stream()), "mnrf_ray_fan_backward_n")
        return (out,) + (None,) * 7


class ReflectFn(torch.autograd.Function):
    """mnrf_reflect_compact / mnrf_reflect_backward (train.py:192-252): reflected-ray construction and
    order-preserving compaction.  apply(rays, x_surface, normal, mask, compact) -> (sec (M,8), index (M) int32
    [empty when not compacted], reflect_dir (N,3) [not differentiable: visualisation output]).
    static=True (round 5): no host read of the count -- sec / index keep their CAPACITY of N rows and a fourth output, count
    (device int32 (1,)), says how many exist; the consumers take it as their live row count (n_live; the `_n` entry points).
    n_live: the live rows of the INPUT rays (a second bounce)."""

    @staticmethod
    def forward(ctx, rays, x_surface, normal, mask, compact, static=False, n_live=None):
        rays, x_surface, normal = _c(rays.float()), _c(x_surface.float()), _c(normal.float())
        N = rays.shape[0]
        dev = rays.device
        sec = torch.empty(N, 8, dtype=torch.float32, device=dev)
        index = torch.empty(N, dtype=torch.int32, device=dev)
        count = torch.empty(1, dtype=torch.int32, device=dev)      # (always written by the kernel)
        rdir = torch.empty(N, 3, dtype=torch.float32, device=dev)
        slot = torch.empty(N, dtype=torch.int32, device=dev) if static else None      # inverse of `index` (the gather-form blend)
        p = _lib.ptr
        _lib.check(_lib.lib().mnrf_reflect_compact_n(
            p(rays), p(x_surface), p(normal), None, 0.0, p(_c(mask.float())) if mask is not None else None, N,
            int(bool(compact)), 0.1, p(sec), p(index), p(count), p(rdir), p(n_live), p(slot), _lib.stream()), "mnrf_reflect_compact")
        ctx.static = bool(static)
        ctx.n_live = n_live
        if static:
            ctx.count = count
            ctx.slot = slot      # (not an autograd output: recursion._static_level picks it up from the count tensor)
            count._mnrf_slot = slot
        else:
            M = int(count.item()) if compact else N     # the one host sync per level (train.py:175 does the same)
            sec, index = sec[:M].contiguous(), index[:M].contiguous()
            ctx.count = None
        ctx.save_for_backward(rays, normal, index)
        ctx.compact = bool(compact)
        ctx.set_materialize_grads(False)     # (index / reflect_dir / count carry no gradient: no tensors of zeros made for them)
        ctx.mark_non_differentiable(index, rdir, count)
        return sec, index, rdir, count

    @staticmethod
    def backward(ctx, g_sec, _gi, _gr, _gc):
        rays, normal, index = ctx.saved_tensors
        N, M = rays.shape[0], index.shape[0]
        dev = rays.device
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
        g_xs, g_n, g_rays = f(N, 3), f(N, 3), f(N, 8)
        p = _lib.ptr
        if g_sec is None:
            return torch.zeros_like(rays), torch.zeros(N, 3, device=dev), torch.zeros(N, 3, device=dev), None, None, None, None
        if ctx.static:     # gather form through the inverse index (identity when not compacted): one launch, nothing zero-filled in front
            _lib.check(_lib.lib().mnrf_reflect_backward_gather_n(
                p(rays), p(normal), p(ctx.slot), p(g_sec.contiguous().float()), N, p(g_xs), p(g_n), p(g_rays), p(ctx.n_live),
                _lib.stream()), "mnrf_reflect_backward_gather_n")
        else:
            _lib.check(_lib.lib().mnrf_reflect_backward_n(
                p(rays), p(normal), p(index) if ctx.compact else None, M, p(g_sec.contiguous().float()), N, p(g_xs), p(g_n),
                p(g_rays), p(ctx.count), _lib.stream()), "mnrf_reflect_backward")
        return g_rays, g_xs, g_n, None, None, None, None


class Blend2Fn(torch.autograd.Function):
    """Both blends of a recursion level (rgb_coarse and rgb_fine, train.py:263-296) as ONE launch forward and ONE backward
    (mnrf_blend2_n / mnrf_blend2_backward_n, gather form through the compaction's inverse index `slot`): the static training route's
    blend.  apply(base_a, sec_a, base_b, sec_b, slot, mask, detach_sec, n_live) -> (out_a, out_b); tensor b may be None."""

    @staticmethod
    def forward(ctx, base_a, sec_a, base_b, sec_b, slot, mask, detach_sec=False, n_live=None):
        base_a, sec_a, mask = _c(base_a.float()), _c(sec_a.float()), _c(mask.float())
        base_b = None if base_b is None else _c(base_b.float())
        sec_b = None if sec_b is None else _c(sec_b.float())
        N = base_a.shape[0]
        c = base_a.shape[1] if base_a.dim() == 2 else 1
        out_a = torch.empty_like(base_a)
        out_b = None if base_b is None else torch.empty_like(base_b)
        p = _lib.ptr
        _lib.check(_lib.lib().mnrf_blend2_n(p(base_a), p(sec_a), p(base_b), p(sec_b), p(slot), p(mask), N, c, p(out_a), p(out_b), p(n_live),
                                            _lib.stream()), "mnrf_blend2_n")
        ctx.save_for_backward(mask, slot)
        ctx.c, ctx.detach_sec, ctx.n_live = c, bool(detach_sec), n_live
        ctx.sec_shapes = (tuple(sec_a.shape), None if sec_b is None else tuple(sec_b.shape))
        ctx.set_materialize_grads(False)
        if out_b is None:
            out_b = torch.empty(0, device=base_a.device)
            ctx.mark_non_differentiable(out_b)
        return out_a, out_b

    @staticmethod
    def backward(ctx, g_a, g_b):
        mask, slot = ctx.saved_tensors
        N = mask.shape[0]
        dev = mask.device
        g_a = None if g_a is None else g_a.contiguous().float()
        g_b = None if (g_b is None or ctx.sec_shapes[1] is None) else g_b.contiguous().float()
        gb_a = None if g_a is None else torch.empty_like(g_a)
        gb_b = None if g_b is None else torch.empty_like(g_b)
        want_sec = not ctx.detach_sec
        gs_a = torch.empty(ctx.sec_shapes[0], dtype=torch.float32, device=dev) if (want_sec and g_a is not None) else None
        gs_b = torch.empty(ctx.sec_shapes[1], dtype=torch.float32, device=dev) if (want_sec and g_b is not None) else None
        p = _lib.ptr
        if g_a is not None or g_b is not None:
            _lib.check(_lib.lib().mnrf_blend2_backward_n(p(g_a), p(g_b), p(slot), p(mask), N, ctx.c, p(gb_a), p(gs_a), p(gb_b), p(gs_b),
                                                         p(ctx.n_live), _lib.stream()), "mnrf_blend2_backward_n")
        return gb_a, gs_a, gb_b, gs_b, None, None, None, None


class BlendFn(torch.autograd.Function):
    """mnrf_blend_scatter / mnrf_blend_backward (train.py:261-296): out = m*part + (1-m)*base with part = sec
    scattered through index (rows without a source keep base.detach()).  `index` None/empty + M == N: direct."""

    @staticmethod
    def forward(ctx, base, sec, index, mask, compact, detach_sec=False, n_sec_live=None, n_live=None):
        """detach_sec: the reflected colour is a constant of the blend (train.py:284-289, --detach_ref_color_for_blend).
        n_sec_live / n_live: device int32 (1,) or None -- the rows of sec / index, resp. of base / mask, that exist (static step)."""
        base, sec, mask = _c(base.float()), _c(sec.float()), _c(mask.float())
        N = base.shape[0]
        c = base.shape[1] if base.dim() == 2 else 1
        out = torch.empty_like(base)
        p = _lib.ptr
        idx = index if compact else None
        _lib.check(_lib.lib().mnrf_blend_scatter_n(p(base), p(sec), p(idx), sec.shape[0], p(mask), N, c, p(out), None,
                                                   p(n_sec_live), p(n_live), _lib.stream()), "mnrf_blend_scatter")
        ctx.n_sec_live, ctx.n_live = n_sec_live, n_live
        ctx.save_for_backward(mask, index)
        ctx.compact, ctx.c, ctx.m, ctx.detach_sec = bool(compact), c, sec.shape[0], bool(detach_sec)
        return out

    @staticmethod
    def backward(ctx, g_out):
        mask, index = ctx.saved_tensors
        N = mask.shape[0]
        g_out = g_out.contiguous().float()
        g_base = torch.empty_like(g_out)
        g_sec = None if ctx.detach_sec else torch.empty((ctx.m,) + tuple(g_out.shape[1:]), dtype=torch.float32, device=g_out.device)
        p = _lib.ptr
        _lib.check(_lib.lib().mnrf_blend_backward_n(p(g_out), p(mask), p(index) if ctx.compact else None, ctx.m, N, ctx.c,
                                                    p(g_base), p(g_sec), p(ctx.n_sec_live), p(ctx.n_live), _lib.stream()),
                   "mnrf_blend_backward")
        return g_base, g_sec, None, None, None, None, None, None
