"""Whitted-style reflected-ray recursion around render_rays, in both rule sets of the reference:

  * `NeRFSystem.forward` / `render_rays_chunk_recursively`  -- TRAIN semantics, train.py:102-348
  * `batched_inference`                                      -- EVAL semantics, eval.py:114-172,
    293-360, 506-548, 614-740 (core path + roughness; the scene-editing demo branches
    place-mirror / substitution / new-object are out of scope)

Python here is the recursion driver only: mask thresholding, reflected-ray construction,
order-preserving compaction and blending are the HIP kernels mnrf_threshold_mask,
mnrf_reflect_compact and mnrf_blend_scatter.  One 4-byte device->host read per level decides
whether (and how many) reflected rays are traced -- the reference syncs at the same place
through `mirror_mask.bool().any()` (train.py:175, eval.py:315).
"""
import os
from collections import defaultdict
from types import SimpleNamespace

import torch
from torch import nn

from . import _lib
from .mirror_nerf import Embedding, MirrorNeRF
from .rendering import render_rays

RAY_FORWARD_OFFSET = 0.1   # train.py:232, eval.py:529 (absolute near of a reflected ray)
JITTER_RAYS = 262144       # rays per batched group of jittered reflections (roughness, eval.py:622-674)


def _f(dev, *s):
    return torch.empty(*s, dtype=torch.float32, device=dev)


def _threshold_(mask, want_any=True):
    """In place m[m>0.5]=1, m[m<0.5]=0 (exactly 0.5 untouched); returns any(m != 0) as a bool.
    want_any=False skips the device->host read of the flag (a stream sync) when the caller does not branch on it."""
    n = mask.shape[0]
    flag = torch.full((1,), 0, dtype=torch.int32, device=mask.device)      # (not torch.zeros: that is a memset, ~40 us of idle GPU)
    if n:
        _lib.check(_lib.lib().mnrf_threshold_mask(_lib.ptr(mask), n, _lib.ptr(flag), _lib.stream()),
                   "mnrf_threshold_mask")
    return bool(flag.item()) if want_any else False


def _threshold_async(mask, host_flags, slot):
    """_threshold_ without the stream sync: the flag travels to pinned host memory behind the kernel and an event marks its
    arrival; `_flag_ready` waits for THAT event only, so launches queued meanwhile (the next chunk's primary pass) keep
    the GPU busy."""
    flag = torch.full((1,), 0, dtype=torch.int32, device=mask.device)      # (not torch.zeros: that is a memset, ~40 us of idle GPU)
    _lib.check(_lib.lib().mnrf_threshold_mask(_lib.ptr(mask), mask.shape[0], _lib.ptr(flag), _lib.stream()), "mnrf_threshold_mask")
    host = host_flags[slot:slot + 1]
    host.copy_(flag, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return host, ev, flag     # (flag: kept alive until the copy has run)


def _flag_ready(pending):
    host, ev, _ = pending
    ev.synchronize()
    return bool(host.item())


def _reflect(rays, x_surface, normal, mask, compact, normal_noise=None, noise_std=0.0, want_dir=True):
    """-> (secondary rays (M,8), index (M,) int32 or None when not compacted, reflect_dir (N,3))."""
    N = rays.shape[0]
    dev = rays.device
    sec = _f(dev, N, 8)
    index = torch.empty(N, dtype=torch.int32, device=dev)
    count = torch.full((1,), 0, dtype=torch.int32, device=dev)
    rdir = _f(dev, N, 3) if want_dir else None
    p = _lib.ptr
    _lib.check(_lib.lib().mnrf_reflect_compact(
        p(rays), p(x_surface.contiguous()), p(normal.contiguous()), p(normal_noise), float(noise_std),
        p(mask.contiguous()) if mask is not None else None, N, int(bool(compact)), RAY_FORWARD_OFFSET,
        p(sec), p(index), p(count), p(rdir), _lib.stream()), "mnrf_reflect_compact")
    M = int(count.item()) if compact else N
    return sec[:M], (index[:M] if compact else None), rdir


def _blend(base, sec, index, mask, want_reflect):
    """m*part + (1-m)*base with part = sec scattered through index (or sec itself)."""
    N, c = base.shape[0], (base.shape[1] if base.dim() == 2 else 1)
    dev = base.device
    out = torch.empty_like(base)
    refl = torch.empty_like(base) if want_reflect else None
    p = _lib.ptr
    _lib.check(_lib.lib().mnrf_blend_scatter(
        p(base.contiguous()), p(sec.contiguous()), p(index), sec.shape[0], p(mask), N, c, p(out), p(refl),
        _lib.stream()), "mnrf_blend_scatter")
    return out, refl


def _reflect_autograd(rays, x_surface, normal, mask, compact):
    """Reflected rays with gradient history (train.py:205 "not detach() to jointly optimize"): the same HIP
    kernel forward, mnrf_reflect_backward backward."""
    from .autograd import ReflectFn
    sec, index, rdir, _count = ReflectFn.apply(rays, x_surface, normal, mask, bool(compact))
    return sec, (index if compact else None), rdir


def _blend_autograd(base, sec, index, mask, want_reflect, detach_sec=False, n_sec_live=None, n_live=None):
    """train.py:263-296 with gradient history (BlendFn); the optional visualisation output is detached."""
    from .autograd import BlendFn
    compact = index is not None
    idx = index if compact else torch.empty(0, dtype=torch.int32, device=base.device)
    out = BlendFn.apply(base, sec, idx, mask, compact, detach_sec, n_sec_live, n_live)
    refl = None
    if want_reflect:
        refl = torch.zeros_like(base)
        if compact:
            refl[index.long()] = sec.detach()
        else:
            refl = sec.detach()
    return out, refl


def _pick_normal(r, sel):
    """train.py:194-215 / eval.py:338-360 -- composited predicted normal, else composited grad normal.  (The reference keys
    on `pred_normal_*`; `surface_normal_*` exists exactly when that does, and is all the ray-fused eval pass produces.)"""
    if f"surface_normal_{sel}" in r:
        return r[f"surface_normal_{sel}"]
    return r[f"surface_normal_grad_{sel}"]


# ----------------------------------------------------------------------------- train semantics
def render_rays_chunk_recursively(models, embeddings, hp, rays_chunk, mirror_mask_prev, recur_level,
                                  extra_chunk, white_back=False, train_geometry_stage=False):
    """train.py:129-348.

    STATIC STEP (round 5; extra_chunk["_static"], set by training.train_step / GraphedTrainStep): the training route without a
    single device->host read.  The reference decides on the host whether anything is traced (`mirror_mask.bool().any()`,
    train.py:175) and how many rays (`secondary_rays[mask.bool()]`, train.py:248-252: a stream sync in the middle of every
    step).  Here the count of the compacted reflections stays on the device (ReflectFn static=True): the reflected rays keep
    the CAPACITY of the chunk, every launch of the nested level takes the count as its live row count (`_n_live`,
    include/mnrf.h) and the blend reads it too.  Values and gradients are those of the host-driven route: with no mirror
    pixel the nested launches find nothing to do and the blend leaves every ray as it is (m = 0: m*x + (1-m)*base == base).
    Differences: `rgb_*_direct` is then present although the reference would not have traced (nothing reads it), and whether
    the ground-truth mask is valid (train.py:153 `(mask >= 0).all()`) is the caller's statement extra_chunk["_gt_valid"]."""
    static = bool(extra_chunk.get("_static")) and not extra_chunk.get("is_eval", False) and torch.is_grad_enabled()
    n_live = extra_chunk.get("_n_live")
    if static and recur_level == 0 and "_rng_share" not in extra_chunk and hp.trace_secondary_rays and not train_geometry_stage:
        # every level of a static step renders the same capacity of rays: the first one draws the random numbers of all of them
        # (two generator launches per step instead of two per level)
        extra_chunk = dict(extra_chunk, _rng_share=(1 + max(0, hp.max_recursive_level), {}))
    r = render_rays(models, embeddings, rays_chunk, hp.N_samples, hp.use_disp, hp.perturb, hp.noise_std,
                    hp.N_importance, hp.chunk, white_back, compute_normal=hp.trace_secondary_rays,
                    **dict(extra_chunk, _guard=False))        # the range guard is read once, in NeRFSystem.forward
    N = rays_chunk.shape[0]
    dev = rays_chunk.device
    sel = "fine" if (hp.N_importance > 0 and not hp.only_one_field) else "coarse"
    if static and N:
        return _static_level(models, embeddings, hp, rays_chunk, mirror_mask_prev, recur_level, extra_chunk, white_back,
                             train_geometry_stage, r, sel, n_live)

    # -- mirror mask (train.py:153-168)
    gt = extra_chunk["mirror_mask"].float()
    any_mirror = None
    # nothing branches on "any mirror pixel" when this level cannot trace: no device->host reads (stream syncs) then
    can_trace = bool(hp.trace_secondary_rays and (not train_geometry_stage) and recur_level < hp.max_recursive_level)
    if recur_level > 0 or bool((gt < 0).any().item()):
        # the reference thresholds `results[...].detach()` in place: the returned predicted
        # mask is the hard one
        if "mirror_mask_fine" in r:
            mask = r["mirror_mask_fine"]
        elif "mirror_mask_coarse" in r:
            mask = r["mirror_mask_coarse"]
        else:
            mask = torch.zeros(N, device=dev)
        any_mirror = _threshold_(mask.detach(), want_any=can_trace)   # in place on the shared storage, like the reference
        mask = mask.detach()
    else:
        mask = gt.clone().contiguous()
    only_in = hp.only_trace_rays_in_mirrors
    if (not only_in) and recur_level > 0:
        mask = mask * mirror_mask_prev.float()
        any_mirror = None
    assumed_mirror = False
    if any_mirror is None:
        if N and can_trace and only_in and not hp.for_vis:
            # compacted reflections: "any mirror pixel" is "the compaction found a ray" -- ONE device->host read (the count the
            # reflect kernel returns) instead of two stream syncs per level
            any_mirror = assumed_mirror = True
        else:
            any_mirror = bool((mask != 0).any().item()) if (N and can_trace) else False

    # -- trace decision (train.py:170-178)
    trace = bool(hp.trace_secondary_rays and (not train_geometry_stage) and (any_mirror or hp.for_vis))
    if recur_level >= hp.max_recursive_level:
        trace = False
    is_eval = extra_chunk.get("is_eval", False)

    traced = False
    grad_path = torch.is_grad_enabled() and r[f"rgb_{sel}"].requires_grad
    reflect_fn, blend_fn = (_reflect_autograd, _blend_autograd) if grad_path else (_reflect, _blend)
    if grad_path and getattr(hp, "detach_ref_color_for_blend", False) and \
            extra_chunk.get("current_epoch", 0) >= getattr(hp, "train_geometry_stage_end_epoch", 4) + 1:
        # train.py:284-289 (the reference reads Lightning's self.current_epoch; here: extra["current_epoch"], which
        # training_step passes as the same number, train.py:426)
        blend_fn = lambda *a: _blend_autograd(*a, detach_sec=True)  # noqa: E731
    if trace and N:
        nrm = _pick_normal(r, sel)
        if getattr(hp, "detach_normal_in_reflection", False):
            nrm = nrm.detach()
        sec, index, rdir = reflect_fn(rays_chunk, r[f"x_surface_{sel}"], nrm, mask, only_in)
        if sec.shape[0] > 0:
            traced = True
            r2 = render_rays_chunk_recursively(models, embeddings, hp, sec.contiguous(), mask, recur_level + 1,
                                               extra_chunk, white_back, train_geometry_stage)
            for typ in ("coarse", "fine"):                               # train.py:263-311
                if f"rgb_{typ}" in r and f"rgb_{typ}" in r2:
                    r[f"rgb_{typ}_direct"] = r[f"rgb_{typ}"]
                    r[f"rgb_{typ}"], refl = blend_fn(r[f"rgb_{typ}"], r2[f"rgb_{typ}"], index, mask, is_eval)
                    if is_eval:
                        r[f"rgb_{typ}_reflect"] = refl
            if is_eval:                                                  # train.py:312-324
                if only_in:
                    d = torch.zeros_like(r[f"depth_{sel}"])
                    d[index.long()] = r2[f"depth_{sel}"]
                    r[f"depth_{sel}_reflect"] = d
                else:
                    r[f"depth_{sel}_reflect"] = r2[f"depth_{sel}"]
                r["secondary_rays_o"] = r[f"x_surface_{sel}"]
                r["reflect_direction"] = rdir
        elif assumed_mirror:
            trace = False       # the compaction found no mirror pixel: what mask.any() would have said up front
    if not trace and is_eval:                                            # train.py:325-346
        for typ in ("coarse", "fine"):
            if f"rgb_{typ}" in r:
                r[f"rgb_{typ}_reflect"] = torch.zeros_like(r[f"rgb_{typ}"])
                r[f"rgb_{typ}_direct"] = torch.zeros_like(r[f"rgb_{typ}"])
        r[f"depth_{sel}_reflect"] = torch.zeros_like(r[f"depth_{sel}"])
        r["secondary_rays_o"] = torch.zeros_like(r[f"rgb_{sel}"])
        r["reflect_direction"] = torch.zeros_like(r[f"rgb_{sel}"])
    del traced
    return r


def _static_level(models, embeddings, hp, rays_chunk, mirror_mask_prev, recur_level, extra_chunk, white_back,
                  train_geometry_stage, r, sel, n_live):
    """One level of render_rays_chunk_recursively in a static step (see there): train.py:153-296 with the mirror-ray count on
    the device.  `r`: this level's render_rays result; n_live: the live rows of rays_chunk (None at level 0)."""
    from .autograd import ReflectFn
    N = rays_chunk.shape[0]
    dev = rays_chunk.device
    can_trace = bool(hp.trace_secondary_rays and (not train_geometry_stage) and recur_level < hp.max_recursive_level)
    gt = extra_chunk["mirror_mask"].float()
    gt_valid = extra_chunk.get("_gt_valid")
    if gt_valid is None:
        raise RuntimeError("static step: extra['_gt_valid'] must say whether the batch's mirror masks are all valid (train.py:153)")
    if recur_level > 0 and not can_trace:
        return r      # (the reference thresholds this level's predicted mask in place, train.py:165-166, in a dict whose masks nobody reads:
                      #  the caller keeps rgb_* only, train.py:263-296 -- one launch less per step)
    if recur_level > 0 or not gt_valid:                           # train.py:155-166: predicted mask, thresholded in place
        if "mirror_mask_fine" in r:
            mask = r["mirror_mask_fine"]
        elif "mirror_mask_coarse" in r:
            mask = r["mirror_mask_coarse"]
        else:
            mask = torch.zeros(N, device=dev)
        md = mask.detach()
        _lib.check(_lib.lib().mnrf_threshold_mask_n(_lib.ptr(md), N, None, _lib.ptr(n_live), _lib.stream()), "mnrf_threshold_mask")
        mask = md
    else:
        mask = gt.contiguous()       # (never written: no clone needed)
    only_in = hp.only_trace_rays_in_mirrors
    if (not only_in) and recur_level > 0:
        mask = mask * mirror_mask_prev.float()                    # train.py:167-168
    if not can_trace:
        return r
    if not r[f"rgb_{sel}"].requires_grad:
        raise RuntimeError("static step: the render carries no gradient history (use the default route for evaluation)")
    detach_sec = bool(getattr(hp, "detach_ref_color_for_blend", False)) and \
        extra_chunk.get("current_epoch", 0) >= getattr(hp, "train_geometry_stage_end_epoch", 4) + 1      # train.py:284-289
    nrm = _pick_normal(r, sel)
    if getattr(hp, "detach_normal_in_reflection", False):
        nrm = nrm.detach()
    sec, index, _rdir, count = ReflectFn.apply(rays_chunk, r[f"x_surface_{sel}"], nrm, mask, bool(only_in), True, n_live)
    r2 = render_rays_chunk_recursively(models, embeddings, hp, sec, mask, recur_level + 1,
                                       dict(extra_chunk, _n_live=count, _compacted=bool(only_in) or bool(extra_chunk.get("_compacted"))),
                                       white_back, train_geometry_stage)
    # train.py:263-296 for both typs in one launch (gather form through the compaction's inverse index: autograd.Blend2Fn)
    from .autograd import Blend2Fn
    typs = [t for t in ("coarse", "fine") if f"rgb_{t}" in r and f"rgb_{t}" in r2]
    if typs:
        a = typs[0]
        b = typs[1] if len(typs) > 1 else None
        out_a, out_b = Blend2Fn.apply(r[f"rgb_{a}"], r2[f"rgb_{a}"], r[f"rgb_{b}"] if b else None, r2[f"rgb_{b}"] if b else None,
                                      count._mnrf_slot, mask, detach_sec, n_live)
        for t, o in ((a, out_a), (b, out_b)):
            if t:
                r[f"rgb_{t}_direct"] = r[f"rgb_{t}"]
                r[f"rgb_{t}"] = o
    return r


class NeRFSystem(nn.Module):
    """The model-holding part of train.NeRFSystem (train.py:33-127): same attribute names
    (`nerf_coarse`, `nerf_fine`, `embedding_xyz`, `embedding_dir`, `models`, `embeddings`) and the
    same `forward(rays, extra)`.  The Lightning training loop around it is out of scope."""

    def __init__(self, hparams, white_back=False):
        super().__init__()
        self.hparams = hparams if not isinstance(hparams, dict) else SimpleNamespace(**hparams)
        hp = self.hparams
        self.train_geometry_stage = getattr(hp, "train_geometry_stage", False)
        self.white_back = white_back
        if getattr(hp, "model_type", "nerf") == "nerf_tcnn":                       # train.py:67-99
            from .mirror_nerf_tcnn import MirrorNeRFTcnn
            self.embedding_xyz = Embedding(0)
            self.embedding_dir = Embedding(0)
            make = lambda: MirrorNeRFTcnn(encoding="hashgrid", bound=hp.bound, predict_normal=hp.predict_normal,  # noqa: E731
                                          predict_mirror_mask=hp.predict_mirror_mask)
        else:
            self.embedding_xyz = Embedding(hp.N_emb_xyz)
            self.embedding_dir = Embedding(hp.N_emb_dir)
            kw = dict(in_channels_xyz=6 * hp.N_emb_xyz + 3, in_channels_dir=6 * hp.N_emb_dir + 3,
                      predict_normal=hp.predict_normal, predict_mirror_mask=hp.predict_mirror_mask)
            make = lambda: MirrorNeRF(**kw)  # noqa: E731
        self.embeddings = {"xyz": self.embedding_xyz, "dir": self.embedding_dir}
        self.nerf_coarse = make()
        self.models = {"coarse": self.nerf_coarse}
        if hp.N_importance > 0 and not hp.only_one_field:
            self.nerf_fine = make()
            self.models["fine"] = self.nerf_fine

    def forward(self, rays, extra=dict()):
        out = self._forward(rays, extra)
        from .mirror_nerf import check_guard, release_transient
        # (training.train_step passes _guard=False: it reads the sticky flag once, after the backward, for both passes)
        if rays.shape[0] and extra.get("_guard", True) and check_guard(self):   # out of range: the models are on fp32 now
            try:
                out = self._forward(rays, extra)
            finally:
                release_transient(self)       # (a range-only trip: for this call only)
        return out

    def _forward(self, rays, extra):
        from .weights import validated
        with validated(self.models.values()):      # (the packed weight images are checked once per call, not per evaluation)
            return self._forward_chunks(rays, extra)

    def _forward_chunks(self, rays, extra):
        hp = self.hparams
        results = defaultdict(list)
        for i in range(0, rays.shape[0], hp.chunk):
            ex = {k: (v[i:i + hp.chunk] if isinstance(v, torch.Tensor) else v) for k, v in extra.items() if k != "_guard"}
            rc = rays[i:i + hp.chunk].contiguous()
            # (train.py:115-119 hands an all-ones "previous mirror mask" to level 0, where nothing reads it: train.py:167-168 multiplies by it
            #  at recur_level > 0 only -- no tensor made for it)
            out = render_rays_chunk_recursively(self.models, self.embeddings, hp, rc, None, 0, ex,
                                                self.white_back, self.train_geometry_stage)
            for k, v in out.items():
                results[k] += [v]
        # (a single chunk -- every training batch -- needs no copy: torch.cat of one tensor is 27 copy kernels per step)
        return {k: (v[0] if len(v) == 1 else torch.cat(v, 0)) for k, v in results.items()}


# ----------------------------------------------------------------------------- eval semantics
_COPY_STREAMS, _PINNED = {}, {}


def _copy_stream(device):
    s = _COPY_STREAMS.get(device)
    if s is None:
        s = _COPY_STREAMS[device] = torch.cuda.Stream(device=device)
    return s


def _pinned(key, rows, tail, dtype):
    """Pinned host staging buffer for one per-ray map of a frame, kept between frames (pinning 51 MB costs more than copying
    them); grown when a larger frame arrives."""
    k = (key, tail, dtype)
    buf = _PINNED.get(k)
    if buf is None or buf.shape[0] < rows:
        buf = _PINNED[k] = torch.empty((rows,) + tail, dtype=dtype).pin_memory()
    return buf


@torch.no_grad()
def batched_inference(models, embeddings, rays, N_samples, N_importance, use_disp, chunk, **kwargs):
    """eval.batched_inference.  kwargs: args (namespace/dict with predict_normal, only_one_field,
    only_one_field_fine_epoch, max_recursive_level, app_control_mirror_roughness, trace_ray_times),
    trace_secondary_rays, normal_noise_std, test_time, white_back (the reference reads the module
    global `dataset.white_back`), to_cpu (True, the default: eval.py:735-736 moves every value to the CPU; False: keep
    everything on the device; "maps": return only the per-ray maps -- rgb, depth, opacity, mirror mask, normals,
    x_surface, ~100 B/ray instead of ~3 KB/ray of per-sample tensors nobody downstream of eval.py:743-894 reads -- on
    the CPU), batch_jitter (roughness: render the trace_ray_times jittered reflections of a level in groups through one
    recursion call each instead of one by one; default on unless draws are injected),
    _normal_noise (iterator of pre-drawn (n,3) standard-normal tensors, for tests)."""
    args = kwargs.get("args")
    if isinstance(args, dict):
        args = SimpleNamespace(**args)
    for flag in ("app_place_new_mirror", "app_reflection_substitution", "app_reflect_newly_placed_objects"):
        if getattr(args, flag, False):
            raise NotImplementedError(f"{flag}: scene-editing demo branches are out of scope")
    trace_flag = kwargs.get("trace_secondary_rays", False)
    test_time = kwargs.get("test_time", True)
    white_back = kwargs.get("white_back", False)
    noise_std = kwargs.get("normal_noise_std", 0)
    noise_iter = kwargs.get("_normal_noise")
    if noise_iter is not None and not isinstance(noise_iter, (list, tuple)):
        # injected draws are kept as a list in kwargs so that a range-guard retry (below) replays the SAME draws
        kwargs = dict(kwargs, _normal_noise=list(noise_iter))
        noise_iter = kwargs["_normal_noise"]
    if noise_iter is not None:
        noise_iter = iter(noise_iter)
    to_cpu = kwargs.get("to_cpu", True)
    # per-ray maps only (the per-sample tensors are neither copied nor, in the final pass, produced at all: render_rays
    # `_maps_only`): what to_cpu="maps" returns anyway; with to_cpu=False it is opt-in (maps_only=True), the full dict stays
    # the default contract
    maps_only = bool(kwargs.get("maps_only", to_cpu == "maps")) and os.environ.get("MNRF_FUSED_EVAL", "1") != "0"
    rough = getattr(args, "app_control_mirror_roughness", False)
    batch_jitter = kwargs.get("batch_jitter", noise_iter is None)   # see the roughness branch of recurse()
    one_field = getattr(args, "only_one_field", False)
    fine_epoch = getattr(args, "only_one_field_fine_epoch", 2)
    sel = "fine" if (N_importance > 0 and not one_field) else "coarse"

    def draw(n, dev):
        if noise_iter is not None:
            return next(noise_iter).to(dev).float().contiguous()
        return torch.randn(n, 3, device=dev)

    # Level 0 runs in two stages so that chunk k+1's primary pass can be queued BEFORE the host waits for chunk k's
    # "any mirror pixel" flag (eval.py:303-312 branches on it): the GPU then never idles on that read.  Not used with the
    # roughness jitters, whose random draws would change order against the primary passes.
    def stage_a(rays_chunk, level, host_flags=None, slot=0):
        r = render_rays(models, embeddings, rays_chunk, N_samples, use_disp, 0, 0, N_importance, chunk,
                        white_back, test_time=test_time,
                        compute_normal=trace_flag and (not args.predict_normal),
                        only_one_field=one_field, only_one_field_fine_epoch=fine_epoch,
                        current_epoch=fine_epoch + 1, _guard=False, _maps_only=maps_only)
        r[f"rgb_{sel}_reflect"] = torch.zeros_like(r[f"rgb_{sel}"])
        r[f"depth_{sel}_reflect"] = torch.zeros_like(r[f"depth_{sel}"])
        mask = None
        for key in (f"mirror_mask_{sel}", "mirror_mask_fine", "mirror_mask_coarse"):
            if key in r:
                mask = r[key]
                break
        # in place (eval.py:303-307); at the last level nothing branches on "any mirror pixel": no host read, so the next
        # chunk's launches queue behind this pass without a stream sync
        last = level >= args.max_recursive_level or not trace_flag
        pending, any_mirror = None, False
        if mask is not None:
            if host_flags is not None and not last and rays_chunk.shape[0]:
                pending = _threshold_async(mask, host_flags, slot)
            else:
                any_mirror = _threshold_(mask, want_any=not last)
        return r, rays_chunk, level, mask, any_mirror, pending

    def recurse(rays_chunk, level):
        return stage_b(stage_a(rays_chunk, level))

    def stage_b(state):
        r, rays_chunk, level, mask, any_mirror, pending = state
        if pending is not None:
            any_mirror = _flag_ready(pending)
        N = rays_chunk.shape[0]
        dev = rays_chunk.device
        only_in = not (level < 1)                                         # eval.py:159
        trace = bool(mask is not None and any_mirror and trace_flag)
        if level >= args.max_recursive_level:
            trace = False
        if not trace or N == 0:
            return r
        # `mirror_mask = mirror_mask.bool()` then `.float()` (eval.py:307, 689): an exact 0.5 blends as 1
        mask = (mask != 0).float()
        normal = _pick_normal(r, sel)
        nn0 = draw(N, dev) if rough else None                             # eval.py:506-511
        sec, index, rdir = _reflect(rays_chunk, r[f"x_surface_{sel}"], normal, mask, only_in, nn0, noise_std)
        r["reflect_direction"] = rdir
        if sec.shape[0] > 0:
            r2 = recurse(sec.contiguous(), level + 1)
            if rough:                                                     # eval.py:622-674
                times = args.trace_ray_times
                if batch_jitter and times > 0:
                    # The reference renders the `times` jittered reflections one after the other (run.sh:187: 64 of them,
                    # each a full secondary render).  Rays are independent, so groups of them go through ONE recursion
                    # call (one field launch per pass for the whole group) and the colours are added in the reference's
                    # order.  Groups are sized to JITTER_RAYS rays (~2.5 GB of per-sample tensors at 64+128 samples).
                    # Not used when a test injects the draws: their order interleaves with the nested levels' draws.
                    mj = max(1, r2[f"rgb_{sel}"].shape[0])     # rows the additions below need (eval.py:663-666)
                    g = 0
                    while g < times:
                        n_g = max(1, min(times - g, JITTER_RAYS // mj))
                        secs = [_reflect(rays_chunk, r[f"x_surface_{sel}"], normal, mask, True, draw(N, dev), noise_std,
                                         want_dir=False)[0] for _ in range(n_g)]
                        r3 = recurse(torch.cat(secs, 0).contiguous(), level + 1)
                        for typ in ("coarse", "fine"):
                            if f"rgb_{typ}" in r2:
                                for piece in r3[f"rgb_{typ}"].view(n_g, -1, 3).unbind(0):
                                    r2[f"rgb_{typ}"] = r2[f"rgb_{typ}"] + piece      # same order of additions as eval.py:663-666
                        g += n_g
                else:
                    for _ in range(times):
                        s2, _, _ = _reflect(rays_chunk, r[f"x_surface_{sel}"], normal, mask, True, draw(N, dev),
                                            noise_std, want_dir=False)
                        r3 = recurse(s2.contiguous(), level + 1)
                        for typ in ("coarse", "fine"):
                            if f"rgb_{typ}" in r2:
                                # the reference adds tensors of M = sum(mask) rows to the first secondary
                                # render; at level 0 that only works when every ray is a mirror
                                r2[f"rgb_{typ}"] = r2[f"rgb_{typ}"] + r3[f"rgb_{typ}"]
                for typ in ("coarse", "fine"):
                    if f"rgb_{typ}" in r2:
                        r2[f"rgb_{typ}"] = r2[f"rgb_{typ}"] / (times + 1)
            r[f"rgb_{sel}"], refl = _blend(r[f"rgb_{sel}"], r2[f"rgb_{sel}"], index, mask, True)
            r[f"rgb_{sel}_reflect"] = refl
            if only_in:
                d = torch.zeros_like(r[f"depth_{sel}"])
                d[index.long()] = r2[f"depth_{sel}"]
                r[f"depth_{sel}_reflect"] = d
            else:
                r[f"depth_{sel}_reflect"] = r2[f"depth_{sel}"]
        return r

    # MNRF_EVAL_PIPELINE (default on): chunk k+1's primary pass is queued before chunk k's reflected pass, so the
    # per-sample tensors of TWO chunks are alive at once (~2 x 3 KB/ray x chunk: 200 MB at chunk 32768) -- set it to 0 on a
    # memory-tight device.  With random roughness draws (no injection) a guard retry draws afresh, as any re-render would.
    results = defaultdict(list)
    # to_cpu="maps" on the GPU: the per-ray maps of a chunk travel to PINNED staging buffers on a side stream while the next
    # chunk renders (pageable destinations made every copy synchronous: 33 ms per 51 MB frame); one host-side copy out of
    # the staging buffers at the end hands back ordinary tensors the caller owns
    stage_maps = to_cpu == "maps" and rays.is_cuda
    staged, staged_rows, hold = {}, defaultdict(int), []
    copy_stream = _copy_stream(rays.device) if stage_maps else None
    starts = list(range(0, rays.shape[0], chunk))
    pipelined = rays.is_cuda and not rough and len(starts) > 1 and os.environ.get("MNRF_EVAL_PIPELINE", "1") != "0"
    host_flags = torch.zeros(2, dtype=torch.int32).pin_memory() if pipelined else None
    ahead = stage_a(rays[:chunk].contiguous(), 0, host_flags, 0) if (pipelined and starts) else None
    for n_c, i in enumerate(starts):
        if pipelined:
            state = ahead
            ahead = stage_a(rays[starts[n_c + 1]:starts[n_c + 1] + chunk].contiguous(), 0, host_flags, (n_c + 1) % 2) \
                if n_c + 1 < len(starts) else None
            out = stage_b(state)
        else:
            out = recurse(rays[i:i + chunk].contiguous(), 0)
        if stage_maps:
            done = torch.cuda.Event()
            done.record()
            copy_stream.wait_event(done)
        for k, v in out.items():
            if to_cpu == "maps" or maps_only:
                if v.dim() <= 2 and (v.dim() == 1 or v.shape[1] <= 3):     # per-ray maps only
                    if stage_maps:
                        buf = staged.get(k)
                        if buf is None:
                            buf = staged[k] = _pinned(k, rays.shape[0], tuple(v.shape[1:]), v.dtype)
                        r0 = staged_rows[k]
                        with torch.cuda.stream(copy_stream):
                            buf[r0:r0 + v.shape[0]].copy_(v, non_blocking=True)
                        staged_rows[k] = r0 + v.shape[0]
                        hold.append(v)      # (alive until the copies have run)
                    else:
                        results[k] += [v.to("cpu", non_blocking=True) if to_cpu == "maps" else (v.cpu() if to_cpu else v)]
            else:
                results[k] += [v.cpu() if to_cpu else v]
    if stage_maps:
        copy_stream.synchronize()
        hold.clear()
        for k, buf in staged.items():
            results[k] = [buf[:staged_rows[k]].clone()]      # out of the (re-used) staging buffer into a tensor of the caller's
    elif to_cpu == "maps" and rays.is_cuda:
        torch.cuda.current_stream().synchronize()
    # range guard of the split arithmetic, once per call (= per frame): a tripped model is on the fp32 kernels now
    from .mirror_nerf import check_guard, release_transient
    if rays.shape[0] and not kwargs.get("_guard_retry") and check_guard([m for m in models.values()]):
        try:
            return batched_inference(models, embeddings, rays, N_samples, N_importance, use_disp, chunk,
                                     **dict(kwargs, _guard_retry=True))
        finally:
            release_transient(list(models.values()))      # (a range-only trip: this frame on fp32, the next one on split again)
    return {k: torch.cat(v, 0) for k, v in results.items()}
