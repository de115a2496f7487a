"""render_rays with the reference's signature and result-dict contract
(models/rendering.py:54-369), executed by the HIP kernels behind include/mnrf.h.

Per call:  mnrf_embed (view encoding, once per ray) -> mnrf_sample_coarse -> mnrf_field_forward
(positions generated in-kernel from rays and depths) -> mnrf_composite -> mnrf_sample_fine ->
mnrf_field_forward -> mnrf_composite.  Python only allocates outputs and assembles the dict.

Differences from the reference that do not change values:
  * `chunk` does not split the MLP evaluation (the kernel is already tiled; results are
    independent of the split, SURVEY 8a noise-floor table);
  * the three detach_* kwargs and `mirror_mask` only steer gradients (rendering.py:223-247, mirror_nerf.py:154-183);
    they are honoured by the backward kernels (flags of mnrf_composite_backward / mnrf_field_backward);
  * random draws can be injected (`_perturb_rand`, `_noise_coarse`, `_noise_fine`, `_u`) so that
    tests can feed the oracle the same numbers; otherwise they come from torch's CUDA generator; `_z_fine` (N, S+N_importance)
    replaces the resampled depths altogether (fixtures G14 of the fine pass);
  * `torch.linspace` tables are built on the CPU like the reference CPU path does and cached;
  * `_n_live` (a device int32 tensor of one element; training only): `rays` is sized for a CAPACITY and only the first
    *_n_live rows exist -- the compacted reflected rays of a training step whose count never visits the host (recursion.py
    "static step"; include/mnrf.h "live row counts on the device").  Every launch is sized for the capacity and leaves the rows past
    the count alone; the result tensors have the capacity's shape, their rows past the count are undefined;
  * `_maps_only=True` (set by batched_inference for to_cpu="maps" / maps_only=True): in eval the final pass runs ray-fused --
    field evaluation + compositing in one kernel -- and the per-sample keys nobody downstream of eval.py:735-736 reads
    (weights_*, pred_normal_*) are not produced; the per-ray maps are identical bit for bit.
Outputs live on rays.device.  When autograd is enabled and a model parameter (or `rays`) requires
grad, the field evaluation and the compositing run through `autograd.FieldFn` / `CompositeFn`,
whose backward passes are HIP kernels too (including the second-order term that reaches the
weights through the normalised density gradient, `normal_*` keys); everything else is unchanged.
"""
import torch

from . import _lib
from .autograd import CompositeFn, EmbedFn, FieldFn
from .mirror_nerf import field_forward
from .weights import params_of

__all__ = ["render_rays", "sample_pdf"]

_LINSPACE = {}


def _linspace01(n, device):
    key = (n, str(device))
    t = _LINSPACE.get(key)
    if t is None:
        t = torch.linspace(0, 1, n).to(device)   # CPU kernel, then copy (SURVEY 8a hazard 2)
        _LINSPACE[key] = t
    return t


def _n_freqs(emb):
    n = getattr(emb, "N_freqs", None)
    if n is None:
        raise RuntimeError("embeddings must expose N_freqs (Embedding modules)")
    return n


def _embed(x, n_freqs, n_live=None):
    x = x.float().contiguous()
    n, c = x.shape
    out = torch.empty(n, c * (2 * n_freqs + 1), dtype=torch.float32, device=x.device)
    if n:
        _lib.check(_lib.lib().mnrf_embed_n(_lib.ptr(x), n, c, n_freqs, _lib.ptr(out), _lib.ptr(n_live), _lib.stream()), "mnrf_embed")
    return out


def sample_pdf(bins_z, weights, N_importance, det=False, u=None, n_live=None):
    """Fine depths for rays whose coarse depths are `bins_z` (N,S) and weights (N,S):
    sample_pdf(mid-points, weights[:,1:-1]) merged with the coarse depths and sorted
    (models/rendering.py:7-51 and 312-326 in one kernel)."""
    N, S = bins_z.shape
    dev = bins_z.device
    if u is None:
        u = _linspace01(N_importance, dev) if det else torch.rand(N, N_importance, device=dev)
    u = u.float().contiguous()
    per_ray = 1 if u.dim() == 2 else 0
    z_fine = torch.empty(N, S + N_importance, dtype=torch.float32, device=dev)
    if N:
        _lib.check(_lib.lib().mnrf_sample_fine_n(_lib.ptr(bins_z), _lib.ptr(weights), N, S, _lib.ptr(u), per_ray,
                                                 N_importance, _lib.ptr(z_fine), _lib.ptr(n_live), _lib.stream()), "mnrf_sample_fine")
    return z_fine


def render_rays(models, embeddings, rays, N_samples=64, use_disp=False, perturb=0, noise_std=1,
                N_importance=0, chunk=1024 * 32, white_back=False, test_time=False, **kwargs):
    L = _lib.lib()
    p = _lib.ptr
    rays = rays.float().contiguous()
    if not rays.is_cuda:
        raise RuntimeError("mirror_nerf_amd.render_rays needs CUDA (ROCm) tensors; there is no CPU path")
    dev = rays.device
    N = rays.shape[0]
    compute_normal = kwargs.get("compute_normal", True)
    n_fx, n_fd = _n_freqs(embeddings["xyz"]), _n_freqs(embeddings["dir"])
    from .mirror_nerf_tcnn import MirrorNeRFTcnn
    hashgrid = isinstance(models["coarse"], MirrorNeRFTcnn)     # BASELINE config 5 (train.py:67-99)
    if hashgrid and (n_fx, n_fd) != (0, 0):
        raise NotImplementedError("the hash-grid field takes raw positions and directions: Embedding(0)/Embedding(0)")
    if not hashgrid:
        for mdl in models.values():
            if (n_fx, n_fd) != (getattr(mdl, "n_freqs_xyz", 10), getattr(mdl, "n_freqs_dir", 4)):
                raise NotImplementedError(f"embeddings Embedding({n_fx}) / Embedding({n_fd}) do not match the model's "
                                          f"{getattr(mdl, 'in_channels_xyz', 63)} / {getattr(mdl, 'in_channels_dir', 27)} input channels")
        n_fd = 4      # the kernel reads 27 view-encoding channels; a model with fewer bands has zero weights on the others
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731

    train = torch.is_grad_enabled() and (rays.requires_grad or any(
        q.requires_grad for mdl in models.values() for q in params_of(mdl)))
    n_live = kwargs.get("_n_live")
    if n_live is not None and (not train or test_time):
        raise NotImplementedError("_n_live (device-side ray count) is a feature of the training path")

    # gradient steering (values are unaffected): rendering.py:223-247 for the compositing weights,
    # mirror_nerf.py:154-183 for what the normal / mirror heads send into geo_feat
    comp_detach, cut_heads, keep_mirror = 0, 0, None
    if train:
        if kwargs.get("detach_density_for_mask_loss", False):
            comp_detach |= _lib.MNRF_DETACH_W_MASK
            cut_heads |= _lib.MNRF_CUT_MIRROR_HEAD
        elif kwargs.get("detach_density_outside_mirror_for_mask_loss", False) and kwargs.get("mirror_mask") is not None:
            mm = kwargs["mirror_mask"]
            gv = kwargs.get("_gt_valid")      # (static step: the caller knows; no device->host read)
            if (gv if gv is not None else not bool((mm < 0).any().item())):
                if mm.shape[0] != N or kwargs.get("_compacted"):
                    # (static route: compacted reflected rays keep the chunk's CAPACITY, so the shapes agree while the rows do
                    #  not -- the level-0 mask would steer the wrong rays without a word.  The count lives on the
                    #  device, so this raises for every compacted level, also in the corner where every ray was a mirror ray.)
                    # the reference indexes the (N,S) weights with this mask and fails the same way when reflected rays
                    # were compacted (train.py:253-259 passes the un-compacted mask on, SURVEY 8a hazard 10)
                    raise IndexError(f"detach_density_outside_mirror_for_mask_loss: mirror_mask has {mm.shape[0]} entries "
                                     f"for {N} rays (compacted reflected rays: use only_trace_rays_in_mirrors=False)")
                keep_mirror = mm.bool().float().contiguous()
        if kwargs.get("detach_density_for_normal_loss", False):
            comp_detach |= _lib.MNRF_DETACH_W_NORMAL
            cut_heads |= _lib.MNRF_CUT_NORMAL_HEAD

    # rendering.py:275-277 -- view encoding once per ray
    # (training, the MirrorNeRF field: encoding and coarse depths come from ONE launch further down, autograd.RayFanFn)
    fan = bool(train and not test_time and not hashgrid and N and N_samples >= 3 and "view_dir" not in kwargs and rays.shape[1] == 8)
    view = kwargs.get("view_dir", rays[:, 3:6])
    if fan:
        dir_emb = None
    elif hashgrid:
        dir_emb = view.float().contiguous()          # Embedding(0): the raw direction (train.py:69-70)
    else:
        dir_emb = EmbedFn.apply(view, n_fd, n_live) if (train and view.requires_grad) else _embed(view.detach(), n_fd, n_live)

    # rendering.py:283-300 -- coarse depths
    z_steps = kwargs.get("_z_steps")
    z_steps = _linspace01(N_samples, dev) if z_steps is None else z_steps.float().contiguous()
    # Training defaults (perturb = noise_std = 1, nothing injected, a coarse and a fine pass): ONE uniform and ONE normal draw per
    # call -- [stratified offsets | inverse-CDF positions] and [coarse | fine density noise] -- instead of five generator launches
    # (rendering.py:189, 296-300; 27 draws its u inside sample_pdf).  The draws are i.i.d. either way.
    S_fine = N_samples + N_importance
    two_pass = train and N and N_importance > 0 and not kwargs.get("only_one_field", False) and kwargs.get("_z_fine") is None
    # `_rng_share` = (k, box): this call is the first of k calls of this shape in one step (the recursion levels of a static
    # training step); it draws for all of them at once and leaves the rest in box["u"] / box["n"], which the later calls take.
    pool_u = pool_n = None
    share, box = kwargs.get("_rng_share") or (1, None)

    def pooled(kind, need, make):
        have = box.get(kind) if box is not None else None
        if have is None or have.numel() < need:
            have = make(need * (share if box is not None else 1))
        if box is not None:
            box[kind] = have[need:]
        return have[:need]
    if two_pass and perturb > 0 and kwargs.get("_perturb_rand") is None and kwargs.get("_u") is None:
        pool_u = pooled("u", N * (N_samples + N_importance), lambda k: torch.rand(k, device=dev))
    auto_noise = {}
    if two_pass and noise_std != 0 and kwargs.get("_noise_coarse") is None and kwargs.get("_noise_fine") is None:
        pool_n = pooled("n", N * (N_samples + S_fine), lambda k: torch.randn(k, device=dev))
        if noise_std != 1:
            pool_n = pool_n * noise_std
        auto_noise = {"_noise_coarse": pool_n[:N * N_samples].view(N, N_samples), "_noise_fine": pool_n[N * N_samples:].view(N, S_fine)}
    prand = None
    if perturb > 0:
        prand = kwargs.get("_perturb_rand")
        if prand is None:
            prand = pool_u[:N * N_samples].view(N, N_samples) if pool_u is not None else torch.rand(N, N_samples, device=dev)
        else:
            prand = prand.float().contiguous()
    # which tensors a pass reads: (rays of the field evaluation, rays of the compositing, view encoding); with the fan, every
    # consumer gets its own view of the rays so that their gradients come back as separate arguments of ONE summing launch
    per_pass = {}
    if fan:
        from .autograd import RayFanFn
        r_cf, r_ff, r_cc, r_fc, de_c, de_f, z_vals = RayFanFn.apply(rays, n_fd, z_steps, N_samples, bool(use_disp), float(perturb),
                                                                   prand, n_live)
        # (argument order of the sum = the order autograd adds in: fine compositing, fine field, coarse compositing, coarse field)
        per_pass = {"_noise_coarse": (r_fc, r_cc, de_c), "_noise_fine": (r_ff, r_cf, de_f)}
    else:
        z_vals = f(N, N_samples)
        if N:
            _lib.check(L.mnrf_sample_coarse_n(p(rays), N, p(z_steps), N_samples, int(bool(use_disp)), float(perturb),
                                              p(prand), p(z_vals), p(n_live), _lib.stream()), "mnrf_sample_coarse")

    has_fine = "fine" in models
    results = {}

    def inference_fused(model, typ, z):
        """Eval, per-ray maps only (`_maps_only`): field evaluation and compositing of the final pass in ONE kernel
        (mnrf_field_composite_fused): a workgroup owns a ray, the four head outputs never leave LDS.  Same maps bit for bit
        as the two-kernel path; the per-sample keys (weights_*, pred_normal_*) are not produced.  Returns False when the
        launch class is not covered (the caller then takes the two-kernel path)."""
        from . import mirror_nerf as _mn
        from .weights import packed_of
        if _mn.precision_of(model) != "split" or z.shape[1] != L.mnrf_fused_samples_per_ray():
            return False
        has_m, has_n = getattr(model, "predict_mirror_mask", True), getattr(model, "predict_normal", True)
        opacity, rgb_map, depth, xs = f(N), f(N, 3), f(N), f(N, 3)
        mask = f(N) if has_m else None
        sn = f(N, 3) if has_n else None
        if _mn.LAUNCH_LOG is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        rc = L.mnrf_field_composite_fused(p(packed_of(model)), N, p(rays), p(z), p(dir_emb), dir_emb.shape[1], int(bool(white_back)),
                                          None, p(opacity), p(rgb_map), p(depth), p(mask), p(sn), p(xs), _lib.stream())
        if rc == -3:          # MNRF_ERR_UNSUPPORTED: the 48-samples-per-wave tuning is switched off
            return False
        _lib.check(rc, "mnrf_field_composite_fused")
        if _mn.LAUNCH_LOG is not None:
            e1.record()
            _mn.LAUNCH_LOG.append((_lib.MNRF_SPLIT_F16 | 0x2000, N * z.shape[1], e0, e1))     # 0x2000: ray-fused launch
        results[f"opacity_{typ}"] = opacity
        results[f"z_vals_{typ}"] = z
        results[f"rgb_{typ}"] = rgb_map
        results[f"depth_{typ}"] = depth
        if has_m:
            results[f"mirror_mask_{typ}"] = mask
        if has_n:
            results[f"surface_normal_{typ}"] = sn
        results[f"_x_surface_{typ}"] = xs
        return True

    def inference(model, typ, z, noise_key):
        """rendering.py:108-264 for one pass.  In the sigma-only coarse pass the reference also
        evaluates (and discards) the normals; they are not computed here."""
        S = z.shape[1]
        sigma_only = typ == "coarse" and test_time and has_fine          # rendering.py:139
        B = N * S
        if train and not sigma_only and N:
            return inference_train(model, typ, z, noise_key)
        if (kwargs.get("_maps_only") and test_time and not sigma_only and N and not hashgrid and not compute_normal
                and noise_std == 0 and kwargs.get(noise_key) is None and inference_fused(model, typ, z)):
            return
        if N and hashgrid:
            o = model.field(B, rays=rays, z_vals=z, spr=S, dirs=dir_emb, sigma_only=sigma_only,
                            grad_normal=compute_normal and not sigma_only)
        elif N:
            o = field_forward(model, B, rays=rays, z_vals=z, spr=S, dir_emb=dir_emb, dir_stride=dir_emb.shape[1],
                              sigma_only=sigma_only, grad_normal=compute_normal and not sigma_only)
        else:
            o = {"sigma": f(0)}
            if not sigma_only:
                o["rgb"] = f(0, 3)
                if getattr(model, "predict_normal", True):
                    o["pred_normal"] = f(0, 3)
                if getattr(model, "predict_mirror_mask", True):
                    o["is_mirror"] = f(0)
                if compute_normal:
                    o["normal"] = f(0, 3)
        noise = kwargs.get(noise_key)
        if noise is not None:
            noise = (noise.float() * noise_std).contiguous()
        elif noise_std != 0:
            noise = torch.randn(N, S, device=dev)                        # rendering.py:189
            if noise_std != 1:      # (x * 1.0 is x: the training default needs no second kernel)
                noise = noise * noise_std
        weights, opacity = f(N, S), f(N)
        full = not sigma_only
        has_m, has_n = full and "is_mirror" in o, full and "pred_normal" in o     # optional heads (mirror_nerf.py:80-99)
        rgb_map = f(N, 3) if full else None
        depth = f(N) if full else None
        mask = f(N) if has_m else None
        sn = f(N, 3) if has_n else None
        sng = f(N, 3) if (full and compute_normal) else None
        nd = f(N) if (has_n and compute_normal) else None
        xs = f(N, 3) if full else None
        if N:
            _lib.check(L.mnrf_composite(
                p(rays), N, S, p(o["sigma"]), p(z), p(noise), p(o.get("rgb")), p(o.get("is_mirror")),
                p(o.get("pred_normal")), p(o.get("normal")) if full else None, int(bool(white_back)),
                p(weights), p(opacity), p(rgb_map), p(depth), p(mask), p(sn), p(sng), p(nd), p(xs),
                _lib.stream()), "mnrf_composite")
        results[f"weights_{typ}"] = weights
        results[f"opacity_{typ}"] = opacity
        results[f"z_vals_{typ}"] = z
        if sigma_only:
            return                                                        # rendering.py:208-209
        results[f"rgb_{typ}"] = rgb_map
        results[f"depth_{typ}"] = depth
        if has_m:
            results[f"mirror_mask_{typ}"] = mask
        if compute_normal:
            results[f"normal_{typ}"] = o["normal"].view(N, S, 3)
            results[f"surface_normal_grad_{typ}"] = sng
        if has_n:
            results[f"pred_normal_{typ}"] = o["pred_normal"].view(N, S, 3)
            results[f"surface_normal_{typ}"] = sn
        if compute_normal and has_n:
            results[f"normal_dif_{typ}"] = nd
        results[f"_x_surface_{typ}"] = xs

    def inference_train(model, typ, z, noise_key):
        """The same pass with autograd-aware kernels (training)."""
        S = z.shape[1]
        rays_f, rays_c, dir_emb_p = per_pass.get(noise_key, (rays, rays, dir_emb))
        if hashgrid:
            from .mirror_nerf_tcnn import TcnnFieldFn
            want = bool(compute_normal)
            if cut_heads or keep_mirror is not None or n_live is not None:
                want = (want, cut_heads, keep_mirror, n_live)
            sigma, rgb, pn, mir, normal, _geo = TcnnFieldFn.apply(model, S, None, rays, z, dir_emb, want,
                                                                  model.encoder.embeddings, *model.mlp_params())
        else:
            want = bool(compute_normal)
            if cut_heads or keep_mirror is not None or n_live is not None:
                want = (want, cut_heads, keep_mirror, n_live)
            sigma, rgb, pn, mir, normal = FieldFn.apply(model, S, None, rays_f, z, dir_emb_p, want, *params_of(model))
        noise = kwargs.get(noise_key)
        if noise is not None:
            noise = (noise.float() * noise_std).contiguous()
        elif noise_key in auto_noise and auto_noise[noise_key].shape == (N, S):
            noise = auto_noise[noise_key]                                 # (already scaled by noise_std)
        elif noise_std != 0:
            noise = torch.randn(N, S, device=dev)
            if noise_std != 1:
                noise = noise * noise_std
        nrm = normal if compute_normal else None
        if not getattr(model, "predict_mirror_mask", True):     # optional heads: evaluated on zero weights, never read
            mir = None
        if not getattr(model, "predict_normal", True):
            pn = None
        fused_u = resample_u() if noise_key == "_noise_coarse" else None
        if fused_u is not None:      # the resampling that follows this pass rides in the compositing launch (mnrf_composite_sample_n)
            weights, opacity, rgb_map, depth, mask, sn, sng, nd, xs, zf = CompositeFn.apply(
                rays_c, sigma.view(N, S), z, noise, rgb, mir, pn, nrm, bool(white_back), comp_detach, keep_mirror, n_live,
                (fused_u, N_importance))
            results["_z_fine_fused"] = zf
        else:
            weights, opacity, rgb_map, depth, mask, sn, sng, nd, xs = CompositeFn.apply(
                rays_c, sigma.view(N, S), z, noise, rgb, mir, pn, nrm, bool(white_back), comp_detach, keep_mirror, n_live)
        results[f"weights_{typ}"] = weights
        results[f"opacity_{typ}"] = opacity
        results[f"z_vals_{typ}"] = z
        results[f"rgb_{typ}"] = rgb_map
        results[f"depth_{typ}"] = depth
        if mir is not None:
            results[f"mirror_mask_{typ}"] = mask
        if compute_normal:
            results[f"normal_{typ}"] = normal.view(N, S, 3)
            results[f"surface_normal_grad_{typ}"] = sng
        if pn is not None:
            results[f"pred_normal_{typ}"] = pn.view(N, S, 3)
            results[f"surface_normal_{typ}"] = sn
        if compute_normal and pn is not None:
            results[f"normal_dif_{typ}"] = nd
        results[f"_x_surface_{typ}"] = xs

    def given_u():
        u = kwargs.get("_u")
        if u is None and perturb == 0:
            u = kwargs.get("_u_det")
        if u is None and pool_u is not None:
            u = pool_u[N * N_samples:].view(N, N_importance)
        return u

    def resample_u():
        """The inverse-CDF positions of the resampling behind the first pass, when it will happen and can ride in that pass's
        compositing launch (training; 3 <= S <= 256, S + N_importance <= 512: mnrf_sample_fine's limits); else None."""
        will = N_importance > 0 and kwargs.get("_z_fine") is None and N and (
            kwargs.get("current_epoch", 0) > kwargs.get("only_one_field_fine_epoch", 2) if kwargs.get("only_one_field", False) else has_fine)
        if not will or not (3 <= N_samples <= 256 and N_samples + N_importance <= 512):
            return None
        u = given_u()
        if u is None:
            u = _linspace01(N_importance, dev) if perturb == 0 else torch.rand(N, N_importance, device=dev)
        return u

    inference(models["coarse"], "coarse", z_vals, "_noise_coarse")

    if N_importance > 0:
        def fine_depths():
            if kwargs.get("_z_fine") is not None:     # tests: the fine depths of another run instead of sample_pdf's
                return kwargs["_z_fine"].float().contiguous()
            if "_z_fine_fused" in results:            # made by the first pass's compositing launch
                return results.pop("_z_fine_fused")
            # weights are detached here, as in the reference (rendering.py:335, 353)
            return sample_pdf(z_vals, results["weights_coarse"].detach(), N_importance, det=(perturb == 0), u=given_u(), n_live=n_live)

        if kwargs.get("only_one_field", False):                           # rendering.py:328-348
            if kwargs.get("current_epoch", 0) > kwargs.get("only_one_field_fine_epoch", 2):
                inference(models["coarse"], "coarse", fine_depths(), "_noise_fine")
        else:                                                             # rendering.py:349-360
            inference(models["fine"], "fine", fine_depths(), "_noise_fine")

    results.pop("_z_fine_fused", None)
    for typ in ("coarse", "fine"):                                        # rendering.py:362-367
        xs = results.pop(f"_x_surface_{typ}", None)
        if f"depth_{typ}" in results:
            results[f"x_surface_{typ}"] = xs
    # range guard of the split arithmetic (mirror_nerf.check_guard): a stand-alone call checks its own launches (one
    # 8-byte device->host read); the recursion drivers pass _guard=False and check once per frame / training forward
    if kwargs.get("_guard", True) and N and not hashgrid:
        from .mirror_nerf import check_guard, release_transient
        if check_guard(list(models.values())):
            try:
                return render_rays(models, embeddings, rays, N_samples, use_disp, perturb, noise_std, N_importance, chunk,
                                   white_back, test_time, **kwargs)
            finally:
                release_transient(list(models.values()))      # (a range-only trip: the models return to the split arithmetic)
    return results
