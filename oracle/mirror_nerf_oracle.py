"""CPU oracle for the Mirror-NeRF volumetric-rendering hot path.

THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it,
and only as the checker.  The product path (`mirror_nerf_amd`) never imports it
and has no CPU fallback.

It is a plain numpy (fp32) restatement, in this repository's own words, of the
algorithm in the reference files

    models/mirror_nerf.py   Embedding (6-38), MirrorNeRF (41-212)
    utils/func.py           l2_normalize (5-7), gradient (10-25)
    models/rendering.py     sample_pdf (7-51), render_rays (54-369)
    train.py                NeRFSystem.forward / render_rays_chunk_recursively (102-348)
    eval.py                 batched_inference core (114-172, 293-360, 506-548, 614-740)
    datasets/ray_utils.py   get_ray_directions / get_rays (6-53)
    metrics.py              psnr (5-15)
    losses.py               ColorLoss, NormalLoss, PlaneConsistentLoss, NormalRegLoss, MirrorMaskLoss, TotalLoss (7-255)

Parity status: PINNED.  `tests/golden/make_golden.py` imports the reference in
the build container, captures inputs/outputs into `tests/golden/*.npz`, and
`tests/test_oracle_golden.py` checks every function here against them
(max-abs <= 2e-6 on composited outputs).  The reference ships no tests or
golden vectors of its own.

Where the reference draws random numbers (stratified jitter, density noise,
random inverse-CDF samples, roughness normals) the oracle takes the drawn
tensors as optional inputs (`_perturb_rand`, `_noise_*`, `_u`, `_normal_noise`)
so that both sides can be fed the same values.
"""
import numpy as np

F32 = np.float32
F64 = np.float64
_EPS32 = F32(np.finfo(np.float32).eps)  # utils/func.py:5


# --------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------
def torch_linspace(start, end, n):
    """torch.linspace(start, end, n) on CPU, bit-for-bit (fp32).

    Used at models/rendering.py:27, 283 and models/mirror_nerf.py:17.  ATen's CPU
    kernel evaluates `start + step*i` for the first half and `end - step*(n-1-i)`
    for the second half, each as one fused multiply-add; `i/(n-1)` differs in
    the last bit for ~10% of the entries (probe).  The fp64
    product of two fp32 numbers is exact, so rounding the fp64 sum once
    reproduces the FMA.
    """
    start = F32(start)
    end = F32(end)
    if n == 1:
        return np.array([start], dtype=F32)
    step = F32((end - start) / F32(n - 1))
    i = np.arange(n, dtype=F64)
    lo = (F64(step) * i + F64(start)).astype(F32)
    hi = (F64(end) - F64(step) * (F64(n - 1) - i)).astype(F32)
    return np.where(np.arange(n) < n // 2, lo, hi).astype(F32)


def l2_normalize(x):
    """utils/func.py:5-7 -- eps clamps the SQUARED norm, inside the sqrt."""
    x = np.asarray(x, dtype=F32)
    sq = np.sum(x * x, axis=-1, keepdims=True, dtype=F32)
    return (x / np.sqrt(np.maximum(sq, _EPS32))).astype(F32)


def _sigmoid(x):
    x = np.asarray(x, dtype=F32)
    return (F32(1) / (F32(1) + np.exp(-x, dtype=F32))).astype(F32)


_SGEMM = None  # optional replacement for the fp32 GEMM  (x, w) -> x @ w.T, see set_sgemm


def set_sgemm(fn):
    """Swap the fp32 matrix product used by every Linear (default: numpy/OpenBLAS).
    bench.py's cpu_baseline leg installs torch's CPU sgemm here -- the library the reference's
    CPU path itself runs on -- so that the baseline is not limited by numpy's BLAS build."""
    global _SGEMM
    _SGEMM = fn


def _linear(x, w, b):
    """nn.Linear: y = x W^T + b with W stored (out, in)."""
    if _SGEMM is not None:
        return (_SGEMM(x, w) + b).astype(F32, copy=False)
    return (x @ w.T + b).astype(F32)


def _cumsum_row(x):
    """torch.cumsum on CPU accumulates fp32 rows in double and rounds every
    partial sum to fp32 (ATen cpu_cum_base_kernel uses acc_type<float>=double)."""
    return np.cumsum(x.astype(F64), axis=-1).astype(F32)


def _cumprod_row(x):
    return np.cumprod(x.astype(F64), axis=-1).astype(F32)


# --------------------------------------------------------------------------
# a1  Embedding                                   models/mirror_nerf.py:6-38
# --------------------------------------------------------------------------
def embedding(x, n_freqs):
    """[x, sin(2^0 x), cos(2^0 x), ..., sin(2^(N-1) x), cos(2^(N-1) x)].

    freq_bands = 2**linspace(0, N-1, N) are exact powers of two, so freq*x is
    exact in fp32 (mirror_nerf.py:17, 34-36).  N = 0 is the identity."""
    x = np.asarray(x, dtype=F32)
    out = [x]
    for k in range(n_freqs):
        f = F32(2.0 ** k)
        out.append(np.sin(f * x, dtype=F32))
        out.append(np.cos(f * x, dtype=F32))
    return np.concatenate(out, axis=-1).astype(F32)


# --------------------------------------------------------------------------
# a2-a5  MirrorNeRF                              models/mirror_nerf.py:41-212
# --------------------------------------------------------------------------
TRUNK_DEPTH = 8
SKIP_AT = 4  # layer index (0-based) whose input is cat[enc_xyz, h]


def field_param_names(predict_normal=True, predict_mirror_mask=True):
    """state_dict key order of the reference module (probe)."""
    names = []
    for i in range(TRUNK_DEPTH):
        names += [f"xyz_encoding_{i+1}.0.weight", f"xyz_encoding_{i+1}.0.bias"]
    names += ["xyz_encoding_final.weight", "xyz_encoding_final.bias"]
    names += ["dir_encoding.0.weight", "dir_encoding.0.bias"]
    names += ["sigma.weight", "sigma.bias", "rgb.0.weight", "rgb.0.bias"]
    if predict_normal:
        names += ["normal_net.0.weight", "normal_net.0.bias",
                  "normal_net.1.weight", "normal_net.1.bias"]
    if predict_mirror_mask:
        names += ["is_mirror_net.0.weight", "is_mirror_net.0.bias",
                  "is_mirror_net.2.weight", "is_mirror_net.2.bias"]
    return names


def forward_density(w, enc_xyz, want_masks=False):
    """mirror_nerf.py:189-197.  Returns sigma (B,1), geo_feat (B,256)[, relu masks]."""
    h = enc_xyz
    masks = []
    for i in range(TRUNK_DEPTH):
        if i == SKIP_AT:
            h = np.concatenate([enc_xyz, h], axis=-1)  # encoding FIRST (192-193)
        pre = _linear(h, w[f"xyz_encoding_{i+1}.0.weight"], w[f"xyz_encoding_{i+1}.0.bias"])
        if want_masks:
            masks.append(pre > 0)
        h = np.maximum(pre, F32(0))
    sigma = _linear(h, w["sigma.weight"], w["sigma.bias"])
    if want_masks:
        return sigma, h, masks
    return sigma, h


def density_gradient(w, xyz, masks, n_freqs):
    """d sigma / d xyz in closed form (SURVEY.md 8a "closed form of the autograd
    normal"); what `gradient(in_xyz, sigma)` evaluates at mirror_nerf.py:136-146
    through utils/func.py:10-25.

    g = w_sigma; for i = 8..1: g = (g * relu_mask_i) @ W_i; at the skip layer the
    first 63 columns go to the encoding gradient; then the encoding Jacobian."""
    n_enc = 3 + 6 * n_freqs
    g = np.broadcast_to(w["sigma.weight"].astype(F32), (xyz.shape[0], w["sigma.weight"].shape[1]))
    g_enc = np.zeros((xyz.shape[0], n_enc), dtype=F32)
    for i in range(TRUNK_DEPTH - 1, -1, -1):
        g = ((g * masks[i]).astype(F32) @ w[f"xyz_encoding_{i+1}.0.weight"]).astype(F32)
        if i == SKIP_AT:
            g_enc = g_enc + g[:, :n_enc]
            g = g[:, n_enc:]
    g_enc = (g_enc + g).astype(F32)
    grad = g_enc[:, 0:3].copy()
    for k in range(n_freqs):
        f = F32(2.0 ** k)
        s = np.sin(f * xyz, dtype=F32)
        c = np.cos(f * xyz, dtype=F32)
        gs = g_enc[:, 3 + 6 * k: 6 + 6 * k]
        gc = g_enc[:, 6 + 6 * k: 9 + 6 * k]
        grad = grad + f * (gs * c - gc * s)
    return grad.astype(F32)


def field_forward(w, x, sigma_only=False, compute_normal=False, n_freqs_xyz=10):
    """MirrorNeRF.forward (mirror_nerf.py:101-187) on numpy arrays.

    x: (B,3) when sigma_only else (B, 3+in_channels_dir) = [raw xyz, embedded dir].
    Returns the same dict keys as the reference: sigma (B,1), geo_feat (B,256),
    pred_normal (B,3) [also when sigma_only -- 154-161 sits outside the
    `if not sigma_only`], rgb (B,3), is_mirror (B,1), normal (B,3)."""
    x = np.asarray(x, dtype=F32)
    out = {}
    xyz = x[:, :3]
    dir_emb = None if sigma_only else x[:, 3:]
    enc = embedding(xyz, n_freqs_xyz)
    if compute_normal:
        sigma, geo, masks = forward_density(w, enc, want_masks=True)
        grad = density_gradient(w, xyz, masks, n_freqs_xyz)
        out["normal"] = l2_normalize(-grad)
    else:
        sigma, geo = forward_density(w, enc)
    out["sigma"] = sigma
    out["geo_feat"] = geo
    if "normal_net.0.weight" in w:  # mirror_nerf.py:85-88 -- no activation between
        hn = _linear(geo, w["normal_net.0.weight"], w["normal_net.0.bias"])
        out["pred_normal"] = l2_normalize(_linear(hn, w["normal_net.1.weight"], w["normal_net.1.bias"]))
    if not sigma_only:
        fin = _linear(geo, w["xyz_encoding_final.weight"], w["xyz_encoding_final.bias"])
        hd = np.maximum(_linear(np.concatenate([fin, dir_emb], -1),
                                w["dir_encoding.0.weight"], w["dir_encoding.0.bias"]), F32(0))
        out["rgb"] = _sigmoid(_linear(hd, w["rgb.0.weight"], w["rgb.0.bias"]))
        if "is_mirror_net.0.weight" in w:  # 256->128 LeakyReLU(0.01) ->1 sigmoid
            hm = _linear(geo, w["is_mirror_net.0.weight"], w["is_mirror_net.0.bias"])
            hm = np.where(hm > 0, hm, F32(0.01) * hm).astype(F32)
            out["is_mirror"] = _sigmoid(_linear(hm, w["is_mirror_net.2.weight"], w["is_mirror_net.2.bias"]))
    return out


# --------------------------------------------------------------------------
# a10  sample_pdf                                 models/rendering.py:7-51
# --------------------------------------------------------------------------
def sample_pdf(bins, weights, n_importance, det=False, eps=1e-5, u=None):
    bins = np.asarray(bins, dtype=F32)
    weights = np.asarray(weights, dtype=F32)
    n_rays, n_s = weights.shape
    eps = F32(eps)
    weights = weights + eps
    pdf = weights / np.sum(weights, axis=-1, keepdims=True, dtype=F32)
    cdf = _cumsum_row(pdf)
    cdf = np.concatenate([np.zeros_like(cdf[:, :1]), cdf], -1)  # (N, n_s+1)
    if u is None:
        if not det:
            raise ValueError("random u must be injected (the oracle owns no RNG)")
        u = torch_linspace(0, 1, n_importance)
    u = np.broadcast_to(np.asarray(u, dtype=F32), (n_rays, n_importance))
    # searchsorted(cdf, u, right=True): number of cdf entries <= u
    inds = np.empty((n_rays, n_importance), dtype=np.int64)
    for r0 in range(0, n_rays, 4096):
        c = cdf[r0:r0 + 4096, None, :]
        inds[r0:r0 + 4096] = np.sum(c <= u[r0:r0 + 4096, :, None], axis=-1)
    below = np.maximum(inds - 1, 0)
    above = np.minimum(inds, n_s)
    cdf_b = np.take_along_axis(cdf, below, 1)
    cdf_a = np.take_along_axis(cdf, above, 1)
    bin_b = np.take_along_axis(bins, below, 1)
    bin_a = np.take_along_axis(bins, above, 1)
    denom = cdf_a - cdf_b
    denom = np.where(denom < eps, F32(1), denom)
    return (bin_b + (u - cdf_b) / denom * (bin_a - bin_b)).astype(F32)


# --------------------------------------------------------------------------
# a6-a11  render_rays                             models/rendering.py:54-369
# --------------------------------------------------------------------------
def _eval_field(w, xyz, dir_emb, n_rays, n_s, sigma_only, compute_normal, n_freqs_xyz, chunk):
    """rendering.py:108-179: flatten, repeat the dir embedding, evaluate in chunks."""
    xyz_ = xyz.reshape(-1, 3)
    outs = {}
    if "encoder.embeddings" in w:      # hash-grid model (train.py:67-99: Embedding(0), so `dir_emb` is the raw direction);
        fwd = lambda x, so: tcnn_field_forward(w, x, w["_cfg"], so, compute_normal)  # noqa: E731  (`_cfg`: hashgrid_config)
    else:
        fwd = lambda x, so: field_forward(w, x, so, compute_normal, n_freqs_xyz)  # noqa: E731
    for i in range(0, xyz_.shape[0], chunk):
        xc = xyz_[i:i + chunk]
        if sigma_only:
            o = fwd(xc, True)
        else:
            ray_idx = np.arange(i, min(i + chunk, xyz_.shape[0])) // n_s
            o = fwd(np.concatenate([xc, dir_emb[ray_idx]], 1), False)
        for k, v in o.items():
            if k != "geo_feat":
                outs.setdefault(k, []).append(v)
    return {k: np.concatenate(v, 0) for k, v in outs.items()}


def _inference(results, w, typ, xyz, z_vals, dir_emb, test_time, has_fine, noise, noise_std,
               white_back, compute_normal, n_freqs_xyz, chunk):
    n_rays, n_s = z_vals.shape
    sigma_only = typ == "coarse" and test_time and has_fine  # rendering.py:139
    o = _eval_field(w, xyz, dir_emb, n_rays, n_s, sigma_only, compute_normal, n_freqs_xyz, chunk)
    sigmas = o["sigma"].reshape(n_rays, n_s)

    deltas = z_vals[:, 1:] - z_vals[:, :-1]
    deltas = np.concatenate([deltas, np.full_like(deltas[:, :1], 1e10)], -1)  # 182-186
    if noise is None:
        noise = np.zeros_like(sigmas)
    noise = (np.asarray(noise, dtype=F32) * F32(noise_std)).astype(F32)  # 189
    alphas = (F32(1) - np.exp(-deltas * np.maximum(sigmas + noise, F32(0)), dtype=F32)).astype(F32)
    shifted = np.concatenate([np.ones_like(alphas[:, :1]), F32(1) - alphas + F32(1e-10)], -1)
    weights = (alphas * _cumprod_row(shifted[:, :-1])).astype(F32)  # 194-199
    wsum = np.sum(weights, axis=-1, dtype=F32)

    results[f"weights_{typ}"] = weights
    results[f"opacity_{typ}"] = wsum
    results[f"z_vals_{typ}"] = z_vals
    if sigma_only:
        return  # 208-209

    rgbs = o["rgb"].reshape(n_rays, n_s, 3)
    rgb_map = np.sum(weights[..., None] * rgbs, axis=1, dtype=F32)
    depth_map = np.sum(weights * z_vals, axis=1, dtype=F32)
    if white_back:
        rgb_map = rgb_map + (F32(1) - wsum[:, None])
    results[f"rgb_{typ}"] = rgb_map.astype(F32)
    results[f"depth_{typ}"] = depth_map.astype(F32)
    if "is_mirror" in o:  # 222-242 (the detach variants do not change values)
        results[f"mirror_mask_{typ}"] = np.sum(weights * o["is_mirror"].reshape(n_rays, n_s), 1, dtype=F32)
    if "normal" in o:  # 246-253
        nrm = o["normal"].reshape(n_rays, n_s, 3)
        results[f"normal_{typ}"] = nrm
        results[f"surface_normal_grad_{typ}"] = np.sum(nrm * weights[..., None], 1, dtype=F32)
    if "pred_normal" in o:  # 254-259
        pn = o["pred_normal"].reshape(n_rays, n_s, 3)
        results[f"pred_normal_{typ}"] = pn
        results[f"surface_normal_{typ}"] = np.sum(pn * weights[..., None], 1, dtype=F32)
    if "normal" in o and "pred_normal" in o:  # 260-264
        dif = np.sum((nrm - pn) ** 2, axis=-1, dtype=F32)
        results[f"normal_dif_{typ}"] = np.sum(weights * dif, 1, dtype=F32)


def render_rays(models, embeddings, rays, N_samples=64, use_disp=False, perturb=0, noise_std=1,
                N_importance=0, chunk=1024 * 32, white_back=False, test_time=False, **kwargs):
    """models: {"coarse": state-dict-of-arrays[, "fine": ...]};
    embeddings: {"xyz": N_freqs_xyz, "dir": N_freqs_dir} (ints).
    Random draws are injected through kwargs (see module docstring)."""
    rays = np.asarray(rays, dtype=F32)
    n_rays = rays.shape[0]
    n_fx, n_fd = embeddings["xyz"], embeddings["dir"]
    compute_normal = kwargs.get("compute_normal", True)
    rays_o, rays_d = rays[:, 0:3], rays[:, 3:6]
    near, far = rays[:, 6:7], rays[:, 7:8]
    dir_emb = embedding(kwargs.get("view_dir", rays_d), n_fd)  # 275-277

    z_steps = kwargs.get("_z_steps")
    if z_steps is None:
        z_steps = torch_linspace(0, 1, N_samples)  # 283
    z_steps = np.asarray(z_steps, dtype=F32)
    if not use_disp:
        z_vals = near * (F32(1) - z_steps) + far * z_steps  # 285
    else:
        z_vals = F32(1) / (F32(1) / near * (F32(1) - z_steps) + F32(1) / far * z_steps)  # 287
    z_vals = np.broadcast_to(z_vals, (n_rays, N_samples)).astype(F32)

    if perturb > 0:  # 291-300
        mid = F32(0.5) * (z_vals[:, :-1] + z_vals[:, 1:])
        upper = np.concatenate([mid, z_vals[:, -1:]], -1)
        lower = np.concatenate([z_vals[:, :1], mid], -1)
        rnd = F32(perturb) * np.asarray(kwargs["_perturb_rand"], dtype=F32)
        z_vals = (lower + (upper - lower) * rnd).astype(F32)

    def positions(z):
        # separate multiply then add (rendering.py:302; hazard 1 of SURVEY 8a)
        return (rays_o[:, None, :] + (rays_d[:, None, :] * z[..., None]).astype(F32)).astype(F32)

    has_fine = "fine" in models
    results = {}
    common = dict(noise_std=noise_std, white_back=white_back, compute_normal=compute_normal,
                  n_freqs_xyz=n_fx, chunk=chunk)
    _inference(results, models["coarse"], "coarse", positions(z_vals), z_vals, dir_emb, test_time,
               has_fine, kwargs.get("_noise_coarse"), **common)

    def fine_points(z, w_mid):  # 312-326
        if kwargs.get("_z_fine") is not None:      # tests: the fine depths of another run (fixtures G14 of the fine pass)
            return np.asarray(kwargs["_z_fine"], dtype=F32)
        mid = F32(0.5) * (z[:, :-1] + z[:, 1:])
        u = kwargs.get("_u")
        if u is None and perturb == 0:
            u = kwargs.get("_u_det")
        znew = sample_pdf(mid, w_mid, N_importance, det=(perturb == 0), u=u)
        return np.sort(np.concatenate([z, znew], -1), -1).astype(F32)

    if N_importance > 0:
        if kwargs.get("only_one_field", False):  # 328-348
            if kwargs.get("current_epoch", 0) > kwargs.get("only_one_field_fine_epoch", 2):
                z_vals = fine_points(z_vals, results["weights_coarse"][:, 1:-1])
                _inference(results, models["coarse"], "coarse", positions(z_vals), z_vals, dir_emb,
                           test_time, has_fine, kwargs.get("_noise_fine"), **common)
        else:  # 349-360
            z_vals = fine_points(z_vals, results["weights_coarse"][:, 1:-1])
            _inference(results, models["fine"], "fine", positions(z_vals), z_vals, dir_emb,
                       test_time, has_fine, kwargs.get("_noise_fine"), **common)

    for typ in ("coarse", "fine"):  # 362-367
        if f"depth_{typ}" in results:
            results[f"x_surface_{typ}"] = (rays_o + rays_d * results[f"depth_{typ}"][:, None]).astype(F32)
    return results


# --------------------------------------------------------------------------
# a12  Whitted recursion, TRAIN semantics                  train.py:102-348
# --------------------------------------------------------------------------
def reflect(rays_d, normal):
    """train.py:217-228 / eval.py:515-523:  r = 2 (w.n) n - w, w = l2n(-d), n = l2n(n)."""
    n = l2_normalize(normal)
    wv = l2_normalize(-np.asarray(rays_d, dtype=F32))
    cos = np.sum(wv * n, axis=-1, dtype=F32)
    return (F32(2) * cos[:, None] * n - wv).astype(F32), n, wv


def _hard_threshold_inplace(m):
    """train.py:165-166 / eval.py:305-306: exactly 0.5 is left untouched."""
    hi = m > F32(0.5)
    lo = m < F32(0.5)
    m[hi] = F32(1)
    m[lo] = F32(0)
    return m


def _pick_normal(r, sel):
    """train.py:194-215 / eval.py:338-360."""
    if f"pred_normal_{sel}" in r:
        return r[f"surface_normal_{sel}"] if f"surface_normal_{sel}" in r else \
            np.sum(r[f"pred_normal_{sel}"] * r[f"weights_{sel}"][..., None], 1, dtype=F32)
    return r[f"surface_normal_grad_{sel}"] if f"surface_normal_grad_{sel}" in r else \
        np.sum(r[f"normal_{sel}"] * r[f"weights_{sel}"][..., None], 1, dtype=F32)


def render_train(models, embeddings, rays, hp, extra, white_back=False, train_geometry_stage=False):
    """NeRFSystem.forward (train.py:102-127) on numpy arrays.

    hp: dict with N_samples, use_disp, perturb, noise_std, N_importance, chunk,
        trace_secondary_rays, only_one_field, max_recursive_level,
        only_trace_rays_in_mirrors, for_vis, (detach flags are value-neutral).
    extra: dict forwarded as **extra_chunk (must hold "mirror_mask" (B,) and "is_eval")."""
    chunk = hp["chunk"]
    pieces = {}
    for i in range(0, rays.shape[0], chunk):
        ex = {k: (v[i:i + chunk] if isinstance(v, np.ndarray) else v) for k, v in extra.items()}
        r = _recurse_train(models, embeddings, rays[i:i + chunk],
                           np.ones(rays[i:i + chunk].shape[0], dtype=bool), 0, hp, ex, white_back,
                           train_geometry_stage)
        for k, v in r.items():
            pieces.setdefault(k, []).append(v)
    return {k: np.concatenate(v, 0) for k, v in pieces.items()}


def _recurse_train(models, embeddings, rays, mask_prev, level, hp, ex, white_back, geo_stage):
    render_kw = {k: v for k, v in ex.items()}
    r = render_rays(models, embeddings, rays, hp["N_samples"], hp["use_disp"], hp["perturb"],
                    hp["noise_std"], hp["N_importance"], hp["chunk"], white_back,
                    compute_normal=hp["trace_secondary_rays"], **render_kw)
    sel = "fine" if (hp["N_importance"] > 0 and not hp.get("only_one_field", False)) else "coarse"

    mask = np.array(ex["mirror_mask"], dtype=F32, copy=True)  # train.py:155
    if (mask < 0).any() or level > 0:
        # `.detach()` aliases the results tensor, so the in-place threshold below
        # also rewrites the returned predicted mask (probe).
        if "mirror_mask_fine" in r:
            mask = r["mirror_mask_fine"]
        elif "mirror_mask_coarse" in r:
            mask = r["mirror_mask_coarse"]
        else:
            mask = np.zeros(rays.shape[0], dtype=F32)
        _hard_threshold_inplace(mask)
    only_in = hp["only_trace_rays_in_mirrors"]
    if (not only_in) and level > 0:
        mask = mask * mask_prev.astype(F32)  # 167-168

    mb = mask.astype(bool)
    trace = bool(hp["trace_secondary_rays"] and (not geo_stage) and (mb.any() or hp.get("for_vis", False)))
    if level >= hp["max_recursive_level"]:
        trace = False

    is_eval = ex.get("is_eval", False)
    if trace:
        far = rays[:, 7:8]
        sec_o = r[f"x_surface_{sel}"]
        refl, _, _ = reflect(rays[:, 3:6], _pick_normal(r, sel))
        sec = np.concatenate([sec_o, refl, np.full_like(far, 0.1), far], -1).astype(F32)  # 230-243
        if only_in:
            sec = sec[mb]
        if sec.shape[0] > 0:
            r2 = _recurse_train(models, embeddings, sec, mask, level + 1, hp, ex, white_back, geo_stage)
            for typ in ("coarse", "fine"):
                if f"rgb_{typ}" in r and f"rgb_{typ}" in r2:
                    r[f"rgb_{typ}_direct"] = r[f"rgb_{typ}"]
                    base = r[f"rgb_{typ}"]
                    if only_in:
                        part = base.copy()
                        part[mb] = r2[f"rgb_{typ}"]
                    else:
                        part = r2[f"rgb_{typ}"]
                    m3 = mask.astype(F32)[:, None]
                    r[f"rgb_{typ}"] = (m3 * part + (F32(1) - m3) * base).astype(F32)  # 289-291
                    if is_eval:
                        if only_in:
                            rr = np.zeros_like(r[f"rgb_{typ}"])
                            rr[mb] = r2[f"rgb_{typ}"]
                            r[f"rgb_{typ}_reflect"] = rr
                        else:
                            r[f"rgb_{typ}_reflect"] = r2[f"rgb_{typ}"]
            if is_eval:
                if only_in:
                    dd = np.zeros_like(r[f"depth_{sel}"])
                    dd[mb] = r2[f"depth_{sel}"]
                    r[f"depth_{sel}_reflect"] = dd
                else:
                    r[f"depth_{sel}_reflect"] = r2[f"depth_{sel}"]
                r["secondary_rays_o"] = sec_o
                r["reflect_direction"] = refl
    else:
        if is_eval:  # 325-346
            for typ in ("coarse", "fine"):
                if f"rgb_{typ}" in r:
                    r[f"rgb_{typ}_reflect"] = np.zeros_like(r[f"rgb_{typ}"])
                    r[f"rgb_{typ}_direct"] = np.zeros_like(r[f"rgb_{typ}"])
            r[f"depth_{sel}_reflect"] = np.zeros_like(r[f"depth_{sel}"])
            r["secondary_rays_o"] = np.zeros_like(r[f"rgb_{sel}"])
            r["reflect_direction"] = np.zeros_like(r[f"rgb_{sel}"])
    return r


# --------------------------------------------------------------------------
# a13/a14  Whitted recursion, EVAL semantics   eval.py:114-172,293-360,506-548,614-740
# --------------------------------------------------------------------------
def render_eval(models, embeddings, rays, N_samples, N_importance, use_disp, chunk, args,
                white_back=False, trace_secondary_rays=True, test_time=True, normal_noise=None):
    """batched_inference core.  args: dict with predict_normal, only_one_field,
    only_one_field_fine_epoch, max_recursive_level, and for the roughness branch
    app_control_mirror_roughness, trace_ray_times, normal_noise_std.
    `normal_noise`: iterator over pre-drawn standard-normal (n,3) arrays, consumed in
    the order the reference calls randn_like (eval.py:508, 627)."""
    pieces = {}
    for i in range(0, rays.shape[0], chunk):
        r = _recurse_eval(models, embeddings, rays[i:i + chunk], 0, N_samples, N_importance,
                          use_disp, chunk, args, white_back, trace_secondary_rays, test_time,
                          normal_noise)
        for k, v in r.items():
            pieces.setdefault(k, []).append(v)
    return {k: np.concatenate(v, 0) for k, v in pieces.items()}


def _recurse_eval(models, embeddings, rays, level, N_samples, N_importance, use_disp, chunk, args,
                  white_back, trace_flag, test_time, normal_noise):
    one_field = args.get("only_one_field", False)
    r = render_rays(models, embeddings, rays, N_samples, use_disp, 0, 0, N_importance, chunk,
                    white_back, test_time=test_time,
                    compute_normal=trace_flag and (not args["predict_normal"]),
                    only_one_field=one_field,
                    only_one_field_fine_epoch=args.get("only_one_field_fine_epoch", 2),
                    current_epoch=args.get("only_one_field_fine_epoch", 2) + 1)
    sel = "fine" if (N_importance > 0 and not one_field) else "coarse"
    only_in = not (level < 1)  # eval.py:159 -- level 0 traces every ray of the chunk
    r[f"rgb_{sel}_reflect"] = np.zeros_like(r[f"rgb_{sel}"])
    r[f"depth_{sel}_reflect"] = np.zeros_like(r[f"depth_{sel}"])

    mask = None
    for key in (f"mirror_mask_{sel}", "mirror_mask_fine", "mirror_mask_coarse"):
        if key in r:
            mask = r[key]
            break
    mb = None
    if mask is not None:
        _hard_threshold_inplace(mask)  # in place on the results tensor (295-307)
        mb = mask.astype(bool)
    trace = bool(mb is not None and mb.any() and trace_flag)
    if level >= args["max_recursive_level"]:
        trace = False
    if not trace:
        return r

    far = rays[:, 7:8]
    sec_o = r[f"x_surface_{sel}"]
    normal = _pick_normal(r, sel)
    rough = args.get("app_control_mirror_roughness", False)
    if rough:  # eval.py:506-511
        normal_bkp = normal.copy()
        normal = normal + (next(normal_noise) * F32(args["normal_noise_std"])).astype(F32)
    refl, _, wv = reflect(rays[:, 3:6], normal)
    r["reflect_direction"] = refl
    near2 = np.full_like(far, 0.1)
    sec = np.concatenate([sec_o, refl, near2, far], -1).astype(F32)
    if only_in:
        sec = sec[mb]
    if sec.shape[0] == 0:
        return r
    rec = dict(N_samples=N_samples, N_importance=N_importance, use_disp=use_disp, chunk=chunk,
               args=args, white_back=white_back, trace_flag=trace_flag, test_time=test_time,
               normal_noise=normal_noise)
    r2 = _recurse_eval(models, embeddings, sec, level + 1, **rec)
    if rough:  # eval.py:622-674 (the reference needs M == N here)
        times = args["trace_ray_times"]
        for _ in range(times):
            nrm = l2_normalize(normal_bkp + (next(normal_noise) * F32(args["normal_noise_std"])).astype(F32))
            rd = (F32(2) * np.sum(wv * nrm, -1, dtype=F32)[:, None] * nrm - wv).astype(F32)
            s2 = np.concatenate([sec_o, rd, near2, far], -1).astype(F32)[mb]
            r3 = _recurse_eval(models, embeddings, s2, level + 1, **rec)
            for typ in ("coarse", "fine"):
                if f"rgb_{typ}" in r2:
                    r2[f"rgb_{typ}"] = r2[f"rgb_{typ}"] + r3[f"rgb_{typ}"]
        for typ in ("coarse", "fine"):
            if f"rgb_{typ}" in r2:
                r2[f"rgb_{typ}"] = (r2[f"rgb_{typ}"] / F32(times + 1)).astype(F32)

    base = r[f"rgb_{sel}"]
    if only_in:
        part = base.copy()
        part[mb] = r2[f"rgb_{sel}"]
    else:
        part = r2[f"rgb_{sel}"]
    m3 = mb.astype(F32)[:, None]
    r[f"rgb_{sel}"] = (m3 * part + (F32(1) - m3) * base).astype(F32)  # 693-697
    if only_in:
        rr = np.zeros_like(r[f"rgb_{sel}"])
        rr[mb] = r2[f"rgb_{sel}"]
        r[f"rgb_{sel}_reflect"] = rr
        dd = np.zeros_like(r[f"depth_{sel}"])
        dd[mb] = r2[f"depth_{sel}"]
        r[f"depth_{sel}_reflect"] = dd
    else:
        r[f"rgb_{sel}_reflect"] = r2[f"rgb_{sel}"]
        r[f"depth_{sel}_reflect"] = r2[f"depth_{sel}"]
    return r


# --------------------------------------------------------------------------
# synthetic pin-hole rays            datasets/ray_utils.py:6-53, blender.py:33-47
# --------------------------------------------------------------------------
def get_ray_directions(H, W, focal):
    """ray_utils.py:6-26 -- pixel (i, j) -> ((i-W/2)/f, -(j-H/2)/f, -1), no +0.5."""
    i, j = np.meshgrid(np.arange(W, dtype=F32), np.arange(H, dtype=F32), indexing="xy")
    f = F32(focal)
    return np.stack([(i - F32(W / 2)) / f, -(j - F32(H / 2)) / f, -np.ones_like(i)], -1).astype(F32)


def get_rays(directions, c2w):
    """ray_utils.py:29-53 -- rotate, normalise, broadcast the camera origin."""
    c2w = np.asarray(c2w, dtype=F32)
    d = (directions @ c2w[:, :3].T).astype(F32)
    d = d / np.linalg.norm(d, axis=-1, keepdims=True).astype(F32)
    o = np.broadcast_to(c2w[:, 3], d.shape)
    return o.reshape(-1, 3).astype(F32), d.reshape(-1, 3).astype(F32)


def look_at_pose(eye=(0.0, -4.0, 1.5), target=(0.0, 0.0, 0.0), up=(0.0, 0.0, 1.0)):
    """Fixed camera-to-world (3,4) used by the synthetic benchmark."""
    eye, target, up = (np.asarray(v, dtype=F64) for v in (eye, target, up))
    zc = eye - target
    zc /= np.linalg.norm(zc)  # camera looks along -z
    xc = np.cross(up, zc)
    xc /= np.linalg.norm(xc)
    yc = np.cross(zc, xc)
    return np.stack([xc, yc, zc, eye], 1).astype(F32)


def synthetic_rays(H, W, near=0.05, far=8.0, camera_angle_x=0.6911112):
    focal = 0.5 * W / np.tan(0.5 * camera_angle_x)  # blender.py:40-42
    o, d = get_rays(get_ray_directions(H, W, focal), look_at_pose())
    nf = np.empty((o.shape[0], 2), dtype=F32)
    nf[:, 0] = near
    nf[:, 1] = far
    return np.concatenate([o, d, nf], 1).astype(F32)


def psnr(pred, gt):
    """metrics.py:5-15."""
    return float(-10.0 * np.log10(np.mean((np.asarray(pred, F64) - np.asarray(gt, F64)) ** 2)))


# --------------------------------------------------------------------------
# losses (losses.py:7-255)
# --------------------------------------------------------------------------
LOSS_DEFAULTS = dict(color_loss_weight=1.0, woMaskRGBtoBlack=False, normal_loss_weight=1e-4,
                     normal_loss_only_inside_mirror=False, normal_reg_loss_weight=0.1, mirror_mask_loss_weight=0.1,
                     model_type="nerf", use_plane_consistent_loss=False, plane_consistent_loss_weight=0.1,
                     smooth_mirror_start_epoch=2, train_mirror_mask_start_epoch=2, train_normal_start_epoch=1)   # opt.py:207-243


def _mean(x):
    """torch's mean of an empty selection is nan (0/0), numpy's warns: restate it explicitly."""
    x = np.asarray(x, F32)
    return F32(np.nan) if x.size == 0 else F32(x.mean(dtype=F32))


def _mse(a, b):
    return _mean((a - b) ** 2)      # nn.MSELoss(reduction="mean") over rows x 3


def color_loss(inputs, batch, hp, train_geometry_stage=False):
    """losses.py:7-51.  In the train_geometry_stage branch with an invalid GT mask (any entry < 0) the reference thresholds
    `inputs["mirror_mask_fine"].detach()` IN PLACE (detach shares storage, losses.py:27-33): the dict entry the mask
    loss reads afterwards holds the thresholded values.  Restated as such: `inputs` is mutated."""
    targets = batch["rgbs"].reshape(-1, 3)
    gt = batch.get("mirror_mask")
    loss = F32(0)
    if train_geometry_stage and gt is not None and (gt < 0).any():
        key = "mirror_mask_fine" if "mirror_mask_fine" in inputs else ("mirror_mask_coarse" if "mirror_mask_coarse" in inputs else None)
        if key is not None:
            m = inputs[key]
            m[m > 0.5] = 1
            m[m < 0.5] = 0
            sel = ~(m != 0)             # .bool(): exactly 0.5 counts as mirror
            for typ in ("coarse", "fine"):
                if f"rgb_{typ}" in inputs:
                    loss = loss + _mse(inputs[f"rgb_{typ}"][sel], targets[sel])
    elif train_geometry_stage and gt is not None and hp["woMaskRGBtoBlack"]:
        sel = ~(gt.reshape(-1) != 0)
        for typ in ("coarse", "fine"):
            if f"rgb_{typ}" in inputs:
                loss = loss + _mse(inputs[f"rgb_{typ}"][sel], targets[sel])
    else:
        for typ in ("coarse", "fine"):
            if f"rgb_{typ}" in inputs:
                loss = loss + _mse(inputs[f"rgb_{typ}"], targets)
    return F32(hp["color_loss_weight"]) * loss


def normal_loss(inputs, batch, hp):
    """losses.py:54-78."""
    gt = batch.get("mirror_mask")
    mm = (gt.reshape(-1) != 0) if (gt is not None and not (gt < 0).any()) else None
    loss = F32(0)
    for typ in ("coarse", "fine"):
        k = f"normal_dif_{typ}"
        if k not in inputs:
            continue
        if mm is not None:
            if not hp["normal_loss_only_inside_mirror"]:
                loss = loss + _mean(inputs[k][~mm])
            loss = loss + _mean(inputs[k][mm]) * F32(100)
        else:
            loss = loss + _mean(inputs[k])
    return F32(hp["normal_loss_weight"]) * loss


def plane_consistent_loss(inputs, batch, hp, plane_idx):
    """losses.py:81-130.  `plane_idx[typ]` (times, 4) are the reference's torch.randint draws, "fine" first
    (4 draws per iteration, in order); rows index the x_surface points INSIDE the GT mirror mask."""
    gt = batch.get("mirror_mask")
    mm = (gt.reshape(-1) != 0) if (gt is not None and not (gt < 0).any()) else None
    loss = F32(0)
    if mm is not None:
        for typ in ("fine", "coarse"):
            k = f"x_surface_{typ}"
            if k not in inputs:
                continue
            pts = inputs[k][mm]
            times = pts.shape[0] // 4
            if times > 0:
                ix = np.asarray(plane_idx[typ]).reshape(times, 4)
                p0, p1, p2, p3 = (pts[ix[:, j]] for j in range(4))
                tri = np.sum(np.cross(p1 - p0, p2 - p0).astype(F32) * (p3 - p0), -1, dtype=F32)
                acc = F32(0)
                for v in np.abs(tri):      # the reference accumulates one iteration at a time
                    acc = F32(acc + v)
                loss = loss + acc / F32(times)
    return F32(hp["plane_consistent_loss_weight"]) * loss


def normal_reg_loss(inputs, batch, hp, ext_supervise_grad_normal=True):
    """losses.py:134-172."""
    rays_d = batch["rays"][..., 3:6].reshape(-1, 3)
    mask = batch["valid_mask"].reshape(-1).astype(bool) if "valid_mask" in batch else np.ones(rays_d.shape[0], bool)
    loss = F32(0)

    def term(nkey, wkey):
        n = inputs[nkey][mask]
        t = np.maximum(n * rays_d[mask][:, None, :], F32(0)).sum(-1, dtype=F32) * inputs[wkey][mask]
        return _mean(t)

    for typ in ("coarse", "fine"):
        if f"pred_normal_{typ}" in inputs:
            loss = loss + term(f"pred_normal_{typ}", f"weights_{typ}")
    if ext_supervise_grad_normal and "normal_fine" in inputs:
        loss = loss + term("normal_fine", "weights_fine")
    return F32(hp["normal_reg_loss_weight"]) * loss


def mirror_mask_loss(inputs, batch, hp):
    """losses.py:175-198.  nn.BCELoss clamps each log at -100; utils/func.py:32-37 (nerf_tcnn) does not."""
    loss = F32(0)
    if "mirror_mask" not in batch:
        return loss
    gt = batch["mirror_mask"].reshape(-1).astype(F32)
    valid = (gt >= 0).astype(F32)
    for typ in ("coarse", "fine"):
        k = f"mirror_mask_{typ}"
        if k not in inputs:
            continue
        p = np.clip(inputs[k].astype(F32), F32(1e-7), F32(1 - 1e-7))
        with np.errstate(divide="ignore"):
            lp, l1p = np.log(p), np.log(F32(1) - p)
        if hp["model_type"] != "nerf_tcnn":
            lp, l1p = np.maximum(lp, F32(-100)), np.maximum(l1p, F32(-100))
        l_ = -(gt * lp + (F32(1) - gt) * l1p)
        loss = loss + _mean(l_ * valid)
    return F32(hp["mirror_mask_loss_weight"]) * loss


def total_loss(inputs, batch, hp=None, train_geometry_stage=False, epoch=-1, plane_idx=None):
    """losses.py:201-255 TotalLoss.forward -> (loss_sum, loss_dict).  `inputs` may be mutated (see color_loss)."""
    h = dict(LOSS_DEFAULTS)
    h.update(hp or {})
    d = {"color_loss": color_loss(inputs, batch, h, train_geometry_stage)}
    if not train_geometry_stage or epoch >= h["train_mirror_mask_start_epoch"]:
        d["mirror_mask_loss"] = mirror_mask_loss(inputs, batch, h)
    if epoch >= h["smooth_mirror_start_epoch"] and h["use_plane_consistent_loss"]:
        d["plane_consistent_loss"] = plane_consistent_loss(inputs, batch, h, plane_idx)
    if not train_geometry_stage or epoch >= h["train_normal_start_epoch"]:
        d["normal_loss"] = normal_loss(inputs, batch, h)
        d["normal_reg_loss"] = normal_reg_loss(inputs, batch, h)
    total = F32(0)
    for v in d.values():
        total = F32(total + v)
    return total, d


# --------------------------------------------------------------------------
# a15  hash-grid field (config 5)                models/mirror_nerf_tcnn.py:13-276
# --------------------------------------------------------------------------
# PARITY UNPINNED for this section: the reference evaluates the multiresolution hash encoding with
# tinycudann (un-vendored, un-pinned, CUDA-only -- README.md:33) and the SH encoding with a CUDA-only
# extension, so neither can run here and the reference holds no vectors for them.  The restatement
# follows the only in-tree statement of the algorithm, models/gridencoder/src/gridencoder.cu
# (fast_hash 51-66, get_grid_index 68-89, kernel_grid 91-272) with the level sizing of
# models/gridencoder/grid.py:181-194 and tinycudann's per_level_scale of mirror_nerf_tcnn.py:38;
# SH from models/shencoder/src/shencoder.cu:49-79 (degree 4); MLPs from mirror_nerf_tcnn.py:52-149,
# 220-259.  It is validated for self-consistency only (HIP kernel vs this file).
HASH_PRIMES = (np.uint32(1), np.uint32(2654435761), np.uint32(805459861))


def hashgrid_config(bound=1.0, n_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19):
    """Level offsets (grid.py:181-194) and log2(per_level_scale) (mirror_nerf_tcnn.py:38)."""
    per_level_scale = np.exp2(np.log2(2048 * bound / n_levels) / (n_levels - 1))
    max_params = 2 ** log2_hashmap_size
    offsets, off = [], 0
    for i in range(n_levels):
        res = int(np.ceil(base_resolution * per_level_scale ** i))
        n = min(max_params, (res + 1) ** 3)
        n = int(np.ceil(n / 8) * 8)
        offsets.append(off)
        off += n
    offsets.append(off)
    return dict(offsets=np.array(offsets, dtype=np.int64), S=float(np.log2(per_level_scale)), H=base_resolution,
                n_levels=n_levels, level_dim=level_dim, bound=float(bound))


def hashgrid_encode(x01, table, cfg, want_grad=False):
    """gridencoder.cu:91-272 (linear interpolation, align_corners=False, gridtype hash).
    x01 (B,3) in [0,1]; table (n_entries, 2) fp32.  Returns (B, 32) [level-major, like tcnn's output]
    and, optionally, d out / d x01 as (B, 32, 3)."""
    x01 = np.asarray(x01, dtype=F32)
    B = x01.shape[0]
    L, C = cfg["n_levels"], cfg["level_dim"]
    out = np.zeros((B, L * C), dtype=F32)
    dydx = np.zeros((B, L * C, 3), dtype=F32) if want_grad else None
    oob = ((x01 < 0) | (x01 > 1)).any(-1)
    for lv in range(L):
        off0, off1 = int(cfg["offsets"][lv]), int(cfg["offsets"][lv + 1])
        hsize = off1 - off0
        # gridencoder.cu:150 (exp2f on the device there); fixed in double precision here and in the kernel's host code
        scale = F32(np.exp2(np.float64(lv) * np.float64(cfg["S"])) * np.float64(cfg["H"]) - 1.0)
        res = np.uint32(np.ceil(scale)) + np.uint32(1)
        pos = x01 * scale + F32(0.5)
        pg = np.floor(pos).astype(np.uint32)
        fr = (pos - pg.astype(F32)).astype(F32)
        acc = np.zeros((B, C), dtype=F32)
        gacc = np.zeros((B, 3, C), dtype=F32)
        for idx in range(8):
            w = np.ones(B, dtype=F32)
            loc = np.empty((B, 3), dtype=np.uint32)
            for d in range(3):
                if idx & (1 << d):
                    w = w * fr[:, d]
                    loc[:, d] = pg[:, d] + np.uint32(1)
                else:
                    w = w * (F32(1) - fr[:, d])
                    loc[:, d] = pg[:, d]
            index = _grid_index(loc, hsize, res)
            val = table[off0 + index]                     # (B, C)
            acc = acc + w[:, None] * val
            if want_grad:
                for gd in range(3):
                    wg = np.full(B, scale, dtype=F32)
                    for d in range(3):
                        if d == gd:
                            continue
                        wg = wg * (fr[:, d] if idx & (1 << d) else (F32(1) - fr[:, d]))
                    sign = F32(1) if idx & (1 << gd) else F32(-1)
                    gacc[:, gd, :] += (sign * wg)[:, None] * val
        acc[oob] = 0
        out[:, lv * C:(lv + 1) * C] = acc
        if want_grad:
            gacc[oob] = 0
            dydx[:, lv * C:(lv + 1) * C, :] = gacc.transpose(0, 2, 1)
    return (out, dydx) if want_grad else out


def _grid_index(loc, hsize, res):
    """get_grid_index (gridencoder.cu:68-89), gridtype hash, align_corners False."""
    stride = np.uint64(1)
    index = np.zeros(loc.shape[0], dtype=np.uint64)
    d = 0
    while d < 3 and stride <= np.uint64(hsize):
        index = index + loc[:, d].astype(np.uint64) * stride
        stride = stride * np.uint64(int(res) + 1)
        d += 1
    index = index.astype(np.uint32)   # the CUDA code does this arithmetic in uint32
    if stride > np.uint64(hsize):
        h = np.zeros(loc.shape[0], dtype=np.uint32)
        for k in range(3):
            h ^= (loc[:, k] * HASH_PRIMES[k]).astype(np.uint32)
        index = h
    return (index % np.uint32(hsize)).astype(np.int64)


def sh4(d):
    """Real spherical harmonics, degree 4 (16 values): shencoder.cu:49-79."""
    d = np.asarray(d, dtype=F32)
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
    o = np.empty((d.shape[0], 16), dtype=F32)
    o[:, 0] = 0.28209479177387814
    o[:, 1] = -0.48860251190291987 * y
    o[:, 2] = 0.48860251190291987 * z
    o[:, 3] = -0.48860251190291987 * x
    o[:, 4] = 1.0925484305920792 * xy
    o[:, 5] = -1.0925484305920792 * yz
    o[:, 6] = 0.94617469575755997 * z2 - 0.31539156525251999
    o[:, 7] = -1.0925484305920792 * xz
    o[:, 8] = 0.54627421529603959 * x2 - 0.54627421529603959 * y2
    o[:, 9] = 0.59004358992664352 * y * (-3.0 * x2 + y2)
    o[:, 10] = 2.8906114426405538 * xy * z
    o[:, 11] = 0.45704579946446572 * y * (1.0 - 5.0 * z2)
    o[:, 12] = 0.3731763325901154 * z * (5.0 * z2 - 3.0)
    o[:, 13] = 0.45704579946446572 * x * (1.0 - 5.0 * z2)
    o[:, 14] = 1.4453057213202769 * z * (x2 - y2)
    o[:, 15] = 0.59004358992664352 * x * (-x2 + 3.0 * y2)
    return o.astype(F32)


def tcnn_field_forward(w, x, cfg, sigma_only=False, compute_normal=False):
    """MirrorNeRFTcnn.forward (mirror_nerf_tcnn.py:151-259).  w: dict with `encoder.embeddings`
    (n,2), `sigma_net.{0,1}.weight`, `color_net.{0,1,2}.weight`, `normal_net.{0,1}.weight`,
    `is_mirror_net.{0,2}.{weight,bias}`.  x: (B,3) or (B,6) = [xyz, raw direction]."""
    x = np.asarray(x, dtype=F32)
    xyz = x[:, :3]
    bound = F32(cfg["bound"])
    x01 = ((xyz + bound) / (F32(2) * bound)).astype(F32)                   # 224
    out = {}
    if compute_normal:
        enc, dydx = hashgrid_encode(x01, w["encoder.embeddings"], cfg, want_grad=True)
    else:
        enc = hashgrid_encode(x01, w["encoder.embeddings"], cfg)
    pre = (enc @ w["sigma_net.0.weight"].T).astype(F32)
    h = (np.maximum(pre, 0) @ w["sigma_net.1.weight"].T).astype(F32)       # 228-233
    sigma, geo = h[:, 0], h[:, 1:]                                         # sigma is NOT rectified here
    out["sigma"] = sigma
    out["geo_feat"] = geo
    if compute_normal:
        g1 = np.broadcast_to(w["sigma_net.1.weight"][0], pre.shape) * (pre > 0)
        genc = (g1.astype(F32) @ w["sigma_net.0.weight"]).astype(F32)       # (B,32)
        grad = np.einsum("bk,bkd->bd", genc, dydx).astype(F32) / (F32(2) * bound)
        out["normal"] = l2_normalize(-grad)
    hn = np.maximum(geo @ w["normal_net.0.weight"].T, 0).astype(F32)       # 249-255
    out["pred_normal"] = l2_normalize((hn @ w["normal_net.1.weight"].T).astype(F32))
    if not sigma_only:
        hc = np.concatenate([sh4(x[:, 3:6]), geo], -1)                     # 238-247
        hc = np.maximum(hc @ w["color_net.0.weight"].T, 0).astype(F32)
        hc = np.maximum(hc @ w["color_net.1.weight"].T, 0).astype(F32)
        out["rgb"] = _sigmoid((hc @ w["color_net.2.weight"].T).astype(F32))
        hm = (geo @ w["is_mirror_net.0.weight"].T + w["is_mirror_net.0.bias"]).astype(F32)
        hm = np.where(hm > 0, hm, F32(0.01) * hm).astype(F32)
        out["is_mirror"] = _sigmoid((hm @ w["is_mirror_net.2.weight"].T + w["is_mirror_net.2.bias"]).astype(F32))
    return out
