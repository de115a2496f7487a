"""MirrorNeRFTcnn with the reference's constructor/forward contract (models/mirror_nerf_tcnn.py:13-276),
evaluated by the fused HIP hash-grid field kernel (csrc/mnrf_tcnn.hip).

The reference builds the multiresolution hash encoding with tinycudann and the view encoding with a
CUDA spherical-harmonics extension; here both live inside one kernel, and the hash table is an
ordinary parameter `encoder.embeddings` of shape (entries, 2) fp32 laid out level after level with
the sizing of models/gridencoder/grid.py:181-194.  tinycudann's private parameter layout is not
reproduced, so tcnn checkpoints do not interchange (and parity against tcnn is unpinned);
the small MLPs keep the reference's names (`sigma_net.N.weight`, `color_net.N.weight`,
`normal_net.N.weight`, `is_mirror_net.{0,2}.{weight,bias}`).

Training: `TcnnFieldFn` chains the forward kernel with `mnrf_tcnn_backward` (table gradient by atomic scatter, MLP
weight gradients by in-kernel fp32 MFMA products, dL/d position and dL/d direction); a gradient arriving at the
density-gradient normal (`normal`) is the second-order term of models/mirror_nerf_tcnn.py:172-218 and is propagated
by a second kernel of the same call (table, sigma_net, position).
"""
import ctypes
import os

import numpy as np
import torch
from torch import nn

from . import _lib


def hashgrid_config(bound=1.0, n_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19):
    per_level_scale = np.exp2(np.log2(2048 * bound / n_levels) / (n_levels - 1))   # mirror_nerf_tcnn.py:38
    max_params = 2 ** log2_hashmap_size
    offsets, off = [], 0
    for i in range(n_levels):                                                       # grid.py:181-194
        res = int(np.ceil(base_resolution * per_level_scale ** i))
        n = int(np.ceil(min(max_params, (res + 1) ** 3) / 8) * 8)
        offsets.append(off)
        off += n
    offsets.append(off)
    return dict(offsets=np.array(offsets, dtype=np.int64), S=float(np.log2(per_level_scale)), H=base_resolution,
                n_levels=n_levels, level_dim=level_dim, bound=float(bound))


# (name, rows, used columns, padded columns) of the weight blob, in blob order; biases follow their matrix
_BLOB = (("sigma_net.0.weight", 64, 32, 32), ("sigma_net.1.weight", 16, 64, 64), ("color_net.0.weight", 64, 31, 32),
         ("color_net.1.weight", 64, 64, 64), ("color_net.2.weight", 3, 64, 64), ("normal_net.0.weight", 64, 15, 16),
         ("normal_net.1.weight", 3, 64, 64), ("is_mirror_net.0.weight", 32, 15, 16), ("is_mirror_net.0.bias", 32, 0, 0),
         ("is_mirror_net.2.weight", 1, 32, 32), ("is_mirror_net.2.bias", 1, 0, 0))


def _offsets17(cfg):
    return (ctypes.c_int64 * 17)(*[int(v) for v in cfg["offsets"]])


GRAD_F16 = os.environ.get("MNRF_TCNN_GRAD_F16", "0") == "1"
# module.mlp_f16 (env MNRF_TCNN_F16=1 sets the default): the MLPs of the forward kernel in single-pass f16 on the matrix pipe
# (one MFMA per product, fp32 accumulation; ~1e-3 relative) instead of hi/lo pairs at fp32 accuracy -- "fp16 MLP on CDNA4 MFMA"
# as BASELINE config 5 words it; the sigma-only launches then run on the matrix pipe too.  The backward recomputes in fp32.
GRAD_FIXED = os.environ.get("MNRF_TCNN_GRAD_FIXED", "1") != "0"
MLP_F16 = os.environ.get("MNRF_TCNN_F16", "0") == "1"
# module.table_f16 (env MNRF_TCNN_TABLE_F16=1 sets the default): the kernels gather from a half2 copy of the table (4 B per entry,
# tinycudann's storage) instead of the fp32 master; see MirrorNeRFTcnn._table
TABLE_F16 = os.environ.get("MNRF_TCNN_TABLE_F16", "0") == "1"
# smallest launch (samples) that encodes level by level into scratch planes (module.enc_planes_min overrides; a huge value
# keeps the one-launch form)
ENC_PLANES_MIN = int(os.environ.get("MNRF_TCNN_PLANES_MIN", "32768"))


def _check_f16_overflow(module):
    """Read the overflow words of earlier packed-f16 backward passes of `module` (include/mnrf.h): a table-gradient sum left the
    f16 range -- it was clamped, not inf, so the optimizer state is intact -- and the module goes back to fp32 atomics."""
    pend = module.__dict__.pop("_mnrf_f16_overflow", None)
    if not pend:
        return False
    hit = False
    for host, ev in pend:
        ev.synchronize()
        hit |= bool(host.item())
    if hit:
        import warnings
        warnings.warn("mirror_nerf_amd: a packed-f16 table-gradient sum left the f16 range in an earlier step (clamped); this model "
                      "accumulates its table gradient with fp32 atomics from now on (table_grad_f16 = False)", RuntimeWarning, stacklevel=3)
        module.table_grad_f16 = False
    return hit


class TcnnFieldFn(torch.autograd.Function):
    """mnrf_tcnn_forward / mnrf_tcnn_backward.
    apply(module, spr, xyz6, rays, z_vals, dirs, want_normal, table, *mlp_params) ->
        sigma (B), rgb (B,3), pred_normal (B,3), is_mirror (B), normal (B,3 or empty), geo_feat (B,15)
        [geo_feat is not differentiable; a gradient arriving at `normal` -- the normalised density gradient -- is the
        second-order term of models/mirror_nerf_tcnn.py:172-218 and is propagated by tcnn_bwd2_kernel]
    Positions/directions come from `xyz6` (B,6) or from rays (N,8) + z_vals (N,spr) with per-ray raw directions
    `dirs` (N,3; None: the ray direction).  `mlp_params`: the 11 tensors of _BLOB in that order."""

    @staticmethod
    def forward(ctx, module, spr, xyz6, rays, z_vals, dirs, want_normal, table, *params):
        # `want_normal` may be a tuple (want_normal, cut_flags, keep_mirror) like FieldFn's: the --detach_density_* options
        # (models/mirror_nerf_tcnn.py:186-215) make normal_net / is_mirror_net see geo_feat.detach()
        ctx.cut, ctx.keep_mirror, ctx.n_live = 0, None, None
        if isinstance(want_normal, tuple):
            wn = want_normal
            want_normal, ctx.cut, km = wn[0], int(wn[1]), wn[2]
            ctx.keep_mirror = None if km is None else km.detach().float().contiguous()
            ctx.n_live = wn[3] if len(wn) > 3 else None      # live rows of `rays` (static training route: recursion.py)
        B = xyz6.shape[0] if xyz6 is not None else rays.shape[0] * spr
        c = lambda t: None if t is None else t.detach().float().contiguous()  # noqa: E731
        xyz6, rays, z_vals, dirs = c(xyz6), c(rays), c(z_vals), c(dirs)
        o = module.field(B, xyz=xyz6, xyz_stride=6, rays=rays, z_vals=z_vals, spr=spr, dirs=dirs,
                         grad_normal=bool(want_normal), want_geo=True, n_live=ctx.n_live)
        ctx.module, ctx.spr, ctx.B = module, spr, B
        ctx.save_for_backward(xyz6, rays, z_vals, dirs)
        ctx.set_materialize_grads(False)
        normal = o.get("normal")
        if normal is None:
            normal = torch.empty(0, 3, dtype=torch.float32, device=o["sigma"].device)
            ctx.mark_non_differentiable(normal)
        ctx.mark_non_differentiable(o["geo_feat"])
        return o["sigma"], o["rgb"], o["pred_normal"], o["is_mirror"], normal, o["geo_feat"]

    @staticmethod
    def backward(ctx, g_sigma, g_rgb, g_pn, g_m, g_normal, _g_geo):
        xyz6, rays, z_vals, dirs = ctx.saved_tensors
        m, B, spr = ctx.module, ctx.B, ctx.spr
        table = m.encoder.embeddings.detach().contiguous()      # the fp32 master: the gradient has its shape
        read_table, tflag = m._table()                          # what the kernels gather from (the half2 copy with table_f16)
        dev = table.device
        c = lambda t: None if t is None else t.contiguous().float()  # noqa: E731
        g_sigma, g_rgb, g_pn, g_m, g_normal = c(g_sigma), c(g_rgb), c(g_pn), c(g_m), c(g_normal)
        need = ctx.needs_input_grad
        d_table = torch.zeros_like(table)
        d_blob = torch.zeros(_lib.lib().mnrf_tcnn_weight_floats(), dtype=torch.float32, device=dev)
        want_x = (xyz6 is not None and need[2]) or (rays is not None and need[3])
        want_d = (xyz6 is not None and need[2]) or (dirs is not None and need[5]) or (dirs is None and rays is not None and need[3])
        # (with a live row count the rows past it are not written: zeros, so that the per-ray sums below stay finite there)
        mk = torch.zeros if ctx.n_live is not None else torch.empty
        d_xyz = mk(B, 3, dtype=torch.float32, device=dev) if want_x else None
        d_dir = mk(B, 3, dtype=torch.float32, device=dev) if want_d else None
        offs = _offsets17(m.cfg)
        # table_grad_f16 (module attribute; env MNRF_TCNN_GRAD_F16=1 sets the default): the big hashed levels accumulate their
        # gradient in half2 with one packed atomic per entry -- tinycudann's gradient precision, 26 % off the step
        flags = ctx.cut | (_lib.MNRF_TCNN_GRAD_F16 if getattr(m, "table_grad_f16", GRAD_F16) else 0)
        # table_grad_fixed (module attribute, default on; MNRF_TCNN_GRAD_FIXED=0 turns it off): one 64-bit integer atomic per table
        # entry -- two 32-bit fixed-point halves under the step's own per-level scale -- instead of two fp32 atomics: exact integer
        # sums (bitwise reproducible), 17 bits below the largest contribution of a level, 4.1 -> 3.0 ms per 1024-ray step
        if not (flags & _lib.MNRF_TCNN_GRAD_F16) and getattr(m, "table_grad_fixed", GRAD_FIXED):
            flags |= _lib.MNRF_TCNN_GRAD_FIXED
            n_copies = _lib.lib().mnrf_tcnn_backward_workspace_floats(offs)
            ws = torch.empty(max(1, _lib.lib().mnrf_tcnn_backward_workspace_floats3(offs, flags, B)), dtype=torch.float32, device=dev)
            ws[:n_copies].zero_()          # (the private copies of the coarse levels; the rest is zeroed / overwritten by the launch)
        else:
            ws = torch.zeros(max(1, _lib.lib().mnrf_tcnn_backward_workspace_floats2(offs, flags)), dtype=torch.float32, device=dev)
        if flags & _lib.MNRF_TCNN_GRAD_F16:
            _check_f16_overflow(m)      # the previous backward's overflow word (its copy finished long ago: no queue drain)
        p = _lib.ptr
        if B:
            _lib.check(_lib.lib().mnrf_tcnn_backward_n(
                read_table.data_ptr(), offs, m.cfg["S"], m.cfg["H"], float(m.bound), p(m._weights()), B, p(xyz6), 6,
                p(rays), p(z_vals), spr, p(dirs), dirs.shape[1] if dirs is not None else 3, p(g_sigma), p(g_rgb), p(g_pn),
                p(g_m), p(g_normal), p(ws), p(d_table), p(d_blob), p(d_xyz), p(d_dir), p(ctx.keep_mirror), flags | tflag,
                p(ctx.n_live), _lib.stream()), "mnrf_tcnn_backward")
        if B and (flags & _lib.MNRF_TCNN_GRAD_F16):
            host = torch.empty(1, dtype=torch.int32, pin_memory=True)
            host.copy_(ws[-4:-3].view(torch.int32), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            m.__dict__.setdefault("_mnrf_f16_overflow", []).append((host, ev))
        grads, off = [], 0
        for _name, rows, used, padded in _BLOB:
            if padded:
                grads.append(d_blob[off:off + rows * padded].view(rows, padded)[:, :used].contiguous())
                off += rows * padded
            else:
                grads.append(d_blob[off:off + rows].clone())
                off += rows
        g_xyz6 = g_rays = g_dirs = None
        if xyz6 is not None and need[2]:
            g_xyz6 = torch.cat([d_xyz, d_dir], 1)
        elif rays is not None:
            N = rays.shape[0]
            per_ray_dir = d_dir.view(N, spr, 3).sum(1) if d_dir is not None else None
            if need[3]:   # x = o + d*z  (rendering.py:302)
                dx = d_xyz.view(N, spr, 3)
                g_rays = torch.zeros_like(rays)
                g_rays[:, 0:3] = dx.sum(1)
                g_rays[:, 3:6] = (dx * z_vals.view(N, spr, 1)).sum(1)
                if dirs is None:
                    g_rays[:, 3:6] += per_ray_dir
            if dirs is not None and need[5]:
                g_dirs = torch.zeros_like(dirs)
                g_dirs[:, :3] = per_ray_dir
        return (None, None, g_xyz6, g_rays, None, g_dirs, None, d_table, *grads)


class _Encoder(nn.Module):
    def __init__(self, n_entries):
        super().__init__()
        self.embeddings = nn.Parameter(torch.empty(n_entries, 2).uniform_(-1e-4, 1e-4))   # grid.py:203-205


class MirrorNeRFTcnn(nn.Module):
    def __init__(self, encoding="hashgrid", encoding_dir="sphere_harmonics", encoding_bg="hashgrid", num_layers=2,
                 hidden_dim=64, geo_feat_dim=15, num_layers_color=3, hidden_dim_color=64, num_layers_bg=2,
                 hidden_dim_bg=64, bound=1, **kwargs):
        super().__init__()
        if (encoding, encoding_dir, num_layers, hidden_dim, geo_feat_dim, num_layers_color, hidden_dim_color) != (
                "hashgrid", "sphere_harmonics", 2, 64, 15, 3, 64) or kwargs.get("bg_radius", 0):
            raise NotImplementedError("the HIP kernel covers the reference configuration of train.py:71-82 only")
        self.bound = bound
        self.cfg = hashgrid_config(bound)
        self.encoder = _Encoder(int(self.cfg["offsets"][-1]))
        self.in_dim = 32
        self.sigma_net = nn.ModuleList([nn.Linear(32, 64, bias=False), nn.Linear(64, 16, bias=False)])
        self.color_net = nn.ModuleList([nn.Linear(31, 64, bias=False), nn.Linear(64, 64, bias=False),
                                        nn.Linear(64, 3, bias=False)])
        self.bg_net = None
        self.predict_normal = kwargs.get("predict_normal", False)
        self.predict_mirror_mask = kwargs.get("predict_mirror_mask", False)
        if not (self.predict_normal and self.predict_mirror_mask):
            raise NotImplementedError("predict_normal and predict_mirror_mask are required (run.sh always sets them)")
        self.normal_net = nn.ModuleList([nn.Linear(15, 64, bias=False), nn.Linear(64, 3, bias=False)])
        self.is_mirror_net = nn.Sequential(nn.Linear(15, 32), nn.LeakyReLU(inplace=True), nn.Linear(32, 1), nn.Sigmoid())
        self._blob = None
        self._blob_key = None

    def mlp_params(self):
        """The 11 tensors of the weight blob, in blob order (include/mnrf.h)."""
        return [self.sigma_net[0].weight, self.sigma_net[1].weight, self.color_net[0].weight, self.color_net[1].weight,
                self.color_net[2].weight, self.normal_net[0].weight, self.normal_net[1].weight, self.is_mirror_net[0].weight,
                self.is_mirror_net[0].bias, self.is_mirror_net[2].weight, self.is_mirror_net[2].bias]

    # ---- weight blob of the small MLPs in the order include/mnrf.h documents
    def _weights(self):
        ps = self.mlp_params()
        from .weights import _GENERATION      # fused optimizers do not bump _version (weights.py)
        key = (_GENERATION[0],) + tuple((p.data_ptr(), p._version) for p in ps)
        if key != self._blob_key:
            import ctypes
            L = _lib.lib()
            src = [p.detach() if (p.dtype == torch.float32 and p.is_contiguous()) else p.detach().float().contiguous() for p in ps]
            blob = torch.empty(L.mnrf_tcnn_weight_floats(), dtype=torch.float32, device=ps[0].device)
            _lib.check(L.mnrf_tcnn_pack_weights((ctypes.c_void_p * 11)(*[t.data_ptr() for t in src]), _lib.ptr(blob), _lib.stream()),
                       "mnrf_tcnn_pack_weights")      # (one launch; it was ten pads, a cat and a pad: 23)
            self._blob = blob
            self._blob_key = key
        return self._blob

    def _table(self):
        """(table pointer tensor, flag): the fp32 table, or -- module.table_f16 (env MNRF_TCNN_TABLE_F16=1 sets the default) -- its
        half2 copy: 4-byte entries, tinycudann's storage (models/mirror_nerf_tcnn.py:39-49 builds a tcnn HashGrid, whose parameters
        live in half precision with an fp32 master copy in the optimizer; SURVEY 8d prices config 5 at 512 B of gathers per sample).
        The fp32 `encoder.embeddings` stays the master the optimizer steps and the gradients land in; the copy is re-made (one launch,
        73 MB of traffic) whenever the master changed."""
        table = self.encoder.embeddings.detach()
        if not getattr(self, "table_f16", TABLE_F16):
            return table.contiguous(), 0
        from .weights import _GENERATION
        key = (_GENERATION[0], table.data_ptr(), self.encoder.embeddings._version)
        if self.__dict__.get("_table_half_key") != key:
            half = self.__dict__.get("_table_half")
            if half is None or half.shape[0] != table.shape[0] or half.device != table.device:
                half = torch.empty(table.shape[0], 2, dtype=torch.float16, device=table.device)
            _lib.check(_lib.lib().mnrf_tcnn_table_half(_lib.ptr(table.contiguous()), table.shape[0], half.data_ptr(), _lib.stream()),
                       "mnrf_tcnn_table_half")
            self.__dict__["_table_half"], self.__dict__["_table_half_key"] = half, key
        return self.__dict__["_table_half"], _lib.MNRF_TCNN_TABLE_F16

    def field(self, B, *, xyz=None, xyz_stride=6, rays=None, z_vals=None, spr=1, dirs=None, sigma_only=False,
              grad_normal=False, want_geo=False, n_live=None):
        """Run the fused kernel; returns flat per-sample tensors like mirror_nerf.field_forward.
        n_live (round 6; ray mode): a device int32 -- B is the capacity, the first *n_live * spr samples exist (mnrf_tcnn_forward_n)."""
        table, tflag = self._table()
        dev = table.device
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
        out = {"sigma": f(B), "pred_normal": f(B, 3)}
        if not sigma_only:
            out["rgb"] = f(B, 3)
            out["is_mirror"] = f(B)
        if grad_normal:
            out["normal"] = f(B, 3)
        if want_geo:
            out["geo_feat"] = f(B, 15)
        flags = (_lib.MNRF_SIGMA_ONLY if sigma_only else 0) | (_lib.MNRF_GRAD_NORMAL if grad_normal else 0) | tflag
        if getattr(self, "mlp_on_valu", False):     # the fp32 VALU kernel instead of hi/lo f16 tiles on the matrix pipe (~1e-6 apart)
            flags |= _lib.MNRF_TCNN_VALU
        elif getattr(self, "mlp_f16", MLP_F16):     # single-pass f16 MLPs: tinycudann's / precision=16's arithmetic (train.py:586)
            flags |= _lib.MNRF_TCNN_F16
        offs = _offsets17(self.cfg)
        p = _lib.ptr
        # level-major encoding planes (include/mnrf.h: enc_workspace): the launches that take the matrix pipe encode level by
        # level into 128 B per sample of scratch -- a third of the fabric traffic of the one-launch form at frame-sized batches
        enc = None
        on_pipe = not grad_normal and not (flags & _lib.MNRF_TCNN_VALU)
        if on_pipe and B >= getattr(self, "enc_planes_min", ENC_PLANES_MIN):
            enc = torch.empty(32 * B, dtype=torch.float32, device=dev)
        from . import mirror_nerf as _mn
        log = _mn.LAUNCH_LOG if B else None
        if log is not None:      # bench.py: kernel time from events on the launching stream
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if B:
            _lib.check(_lib.lib().mnrf_tcnn_forward_n(
                table.data_ptr(), offs, self.cfg["S"], self.cfg["H"], float(self.bound), p(self._weights()), flags, B,
                p(xyz), xyz_stride, p(rays), p(z_vals), spr, p(dirs), dirs.shape[1] if dirs is not None else 3,
                p(out["sigma"]), p(out.get("rgb")), p(out["pred_normal"]), p(out.get("is_mirror")), p(out.get("normal")),
                p(out.get("geo_feat")), p(enc), p(n_live), _lib.stream()), "mnrf_tcnn_forward")
        if log is not None:
            e1.record()
            log.append((flags | 0x1000, B, e0, e1))       # 0x1000: hash-grid field
        return out

    def forward(self, x, compute_normal=True, sigma_only=False, embedding_xyz=None, embedding_dir=None,
                mirror_mask=None, detach_density_outside_mirror_for_mask_loss=False,
                detach_density_for_mask_loss=False, detach_density_for_normal_loss=False):
        """x: (B,3) when sigma_only else (B,6) = [xyz, raw direction] (mirror_nerf_tcnn.py:165-170).
        `sigma` has shape (B,) here, as in the reference (235)."""
        x = x.float().contiguous()
        B = x.shape[0]
        if (not sigma_only and x.shape[1] == 6 and torch.is_grad_enabled()
                and (x.requires_grad or any(q.requires_grad for q in self.parameters()))):
            # training route (geo_feat is a constant of the graph; the density-gradient normal carries its second-order term)
            # the --detach_density_* options (models/mirror_nerf_tcnn.py:186-215): that head sees geo_feat.detach()
            cut, keep = 0, None
            if detach_density_for_normal_loss:
                cut |= _lib.MNRF_CUT_NORMAL_HEAD
            if detach_density_for_mask_loss:
                cut |= _lib.MNRF_CUT_MIRROR_HEAD
            elif detach_density_outside_mirror_for_mask_loss and mirror_mask is not None and not bool((mirror_mask < 0).any()):
                keep = mirror_mask.bool().float().contiguous()          # per sample: inside the mirror the gradient flows
            want = (bool(compute_normal), cut, keep) if (cut or keep is not None) else bool(compute_normal)
            sigma, rgb, pn, mir, normal, geo = TcnnFieldFn.apply(self, 1, x, None, None, None, want,
                                                                 self.encoder.embeddings, *self.mlp_params())
            out = {"normal": normal} if compute_normal else {}
            out.update(sigma=sigma, geo_feat=geo, pred_normal=pn, rgb=rgb, is_mirror=mir.view(B, 1))
            return out
        o = self.field(B, xyz=x, xyz_stride=x.shape[1], sigma_only=sigma_only, grad_normal=compute_normal, want_geo=True)
        out = {}
        if compute_normal:
            out["normal"] = o["normal"]
        out["sigma"] = o["sigma"]
        out["geo_feat"] = o["geo_feat"]
        out["pred_normal"] = o["pred_normal"]
        if not sigma_only:
            out["rgb"] = o["rgb"]
            out["is_mirror"] = o["is_mirror"].view(B, 1)
        return out
