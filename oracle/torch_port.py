"""TEST / BASELINE INFRASTRUCTURE -- never imported by the product (mirror_nerf_amd/).

Plain-torch CPU restatement of the reference's EVAL path for the benchmark workload only
(`eval.batched_inference` with perturb=0, noise_std=0, test_time=True, predict_normal=True:
eval.py:114-172, 293-360, 513-548, 676-740 over models/rendering.py:54-369 and
models/mirror_nerf.py:101-212).  It issues the same kind of torch ops the reference does
(`cat` + `nn.functional.linear` per layer, `cumprod`, `searchsorted`, `sort`, MLP evaluated in
`chunk`-sample pieces), so that `bench.py`'s `cpu_baseline` times what the reference's CPU path
costs on the host cores -- the numpy oracle beside it is a checker first and pays for its
bit-level care (SURVEY 8d "CPU baseline beside it").
Pinned by tests/test_oracle_golden.py::test_torch_port_matches_oracle (max-abs <= 2e-6 on the per-ray maps).
"""
import torch
import torch.nn.functional as F


def _l2n(x):
    """utils/func.py:5-8."""
    return x / torch.sqrt(torch.clamp((x * x).sum(-1, keepdim=True), min=torch.finfo(torch.float32).eps))


def embed(x, n_freqs):
    """models/mirror_nerf.py:20-38."""
    out = [x]
    for k in range(n_freqs):
        f = 2.0 ** k
        out += [torch.sin(f * x), torch.cos(f * x)]
    return torch.cat(out, -1)


def field(w, x, sigma_only):
    """models/mirror_nerf.py:101-212 (compute_normal=False)."""
    xyz = x[:, :3]
    enc = embed(xyz, 10)
    h = enc
    for i in range(8):
        if i == 4:
            h = torch.cat([enc, h], -1)
        h = torch.relu(F.linear(h, w[f"xyz_encoding_{i+1}.0.weight"], w[f"xyz_encoding_{i+1}.0.bias"]))
    out = {"sigma": F.linear(h, w["sigma.weight"], w["sigma.bias"])}
    # the reference evaluates normal_net in the sigma-only pass too (mirror_nerf.py:154-161)
    hn = F.linear(h, w["normal_net.0.weight"], w["normal_net.0.bias"])
    out["pred_normal"] = _l2n(F.linear(hn, w["normal_net.1.weight"], w["normal_net.1.bias"]))
    if sigma_only:
        return out
    fin = F.linear(h, w["xyz_encoding_final.weight"], w["xyz_encoding_final.bias"])
    hd = torch.relu(F.linear(torch.cat([fin, x[:, 3:]], -1), w["dir_encoding.0.weight"], w["dir_encoding.0.bias"]))
    out["rgb"] = torch.sigmoid(F.linear(hd, w["rgb.0.weight"], w["rgb.0.bias"]))
    hm = F.leaky_relu(F.linear(h, w["is_mirror_net.0.weight"], w["is_mirror_net.0.bias"]), 0.01)
    out["is_mirror"] = torch.sigmoid(F.linear(hm, w["is_mirror_net.2.weight"], w["is_mirror_net.2.bias"]))
    return out


def sample_pdf(bins, weights, n_importance):
    """models/rendering.py:7-51 with det=True."""
    weights = weights + 1e-5
    pdf = weights / weights.sum(-1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1)
    u = torch.linspace(0, 1, n_importance).expand(bins.shape[0], n_importance).contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp_min(inds - 1, 0)
    above = torch.clamp_max(inds, weights.shape[1])
    g = torch.stack([below, above], -1).view(bins.shape[0], 2 * n_importance)
    cdf_g = torch.gather(cdf, 1, g).view(-1, n_importance, 2)
    bins_g = torch.gather(bins, 1, g).view(-1, n_importance, 2)
    denom = cdf_g[..., 1] - cdf_g[..., 0]
    denom[denom < 1e-5] = 1
    return bins_g[..., 0] + (u - cdf_g[..., 0]) / denom * (bins_g[..., 1] - bins_g[..., 0])


def _inference(res, w, typ, rays_o, rays_d, z, dir_emb, sigma_only, chunk):
    """models/rendering.py:108-264."""
    n, s = z.shape
    xyz = (rays_o[:, None] + rays_d[:, None] * z[..., None]).view(-1, 3)
    de = None if sigma_only else dir_emb.repeat_interleave(s, 0)
    outs = {}
    for i in range(0, xyz.shape[0], chunk):
        xc = xyz[i:i + chunk] if sigma_only else torch.cat([xyz[i:i + chunk], de[i:i + chunk]], 1)
        for k, v in field(w, xc, sigma_only).items():
            outs.setdefault(k, []).append(v)
    o = {k: torch.cat(v, 0) for k, v in outs.items()}
    sig = o["sigma"].view(n, s)
    deltas = torch.cat([z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], 1e10)], -1)
    alphas = 1 - torch.exp(-deltas * torch.relu(sig))
    shifted = torch.cat([torch.ones_like(alphas[:, :1]), 1 - alphas + 1e-10], -1)
    wts = alphas * torch.cumprod(shifted[:, :-1], -1)
    res[f"weights_{typ}"], res[f"opacity_{typ}"], res[f"z_vals_{typ}"] = wts, wts.sum(1), z
    if sigma_only:
        return
    res[f"rgb_{typ}"] = (wts[..., None] * o["rgb"].view(n, s, 3)).sum(1)
    res[f"depth_{typ}"] = (wts * z).sum(1)
    res[f"mirror_mask_{typ}"] = (wts * o["is_mirror"].view(n, s)).sum(1)
    res[f"surface_normal_{typ}"] = (wts[..., None] * o["pred_normal"].view(n, s, 3)).sum(1)
    res[f"x_surface_{typ}"] = rays_o + rays_d * res[f"depth_{typ}"][:, None]


def render_rays(models, rays, n_samples, n_importance, chunk):
    """models/rendering.py:54-369 at perturb=0, noise_std=0, test_time=True, coarse + fine."""
    rays_o, rays_d, near, far = rays[:, 0:3], rays[:, 3:6], rays[:, 6:7], rays[:, 7:8]
    dir_emb = embed(rays_d, 4)
    t = torch.linspace(0, 1, n_samples)
    z = (near * (1 - t) + far * t).expand(rays.shape[0], n_samples)
    res = {}
    _inference(res, models["coarse"], "coarse", rays_o, rays_d, z, dir_emb, True, chunk)
    mid = 0.5 * (z[:, :-1] + z[:, 1:])
    z_new = sample_pdf(mid, res["weights_coarse"][:, 1:-1], n_importance)
    z_fine = torch.sort(torch.cat([z, z_new], -1), -1)[0]
    _inference(res, models["fine"], "fine", rays_o, rays_d, z_fine, dir_emb, False, chunk)
    return res


@torch.no_grad()
def render_eval(models, rays, n_samples, n_importance, chunk, max_level=1):
    """eval.batched_inference over ray chunks; `models`: {"coarse"/"fine": dict name -> fp32 tensor}."""
    def recurse(rc, level):
        r = render_rays(models, rc, n_samples, n_importance, chunk)
        m = r["mirror_mask_fine"]
        m[m > 0.5] = 1                                               # eval.py:303-307 (in place)
        m[m < 0.5] = 0
        if level >= max_level or not bool(m.bool().any()):
            return r
        n = _l2n(r["surface_normal_fine"])                          # eval.py:515-523
        wv = _l2n(-rc[:, 3:6])
        rdir = 2 * (wv * n).sum(-1, keepdim=True) * n - wv
        sec = torch.cat([r["x_surface_fine"], rdir, torch.full_like(rc[:, 6:7], 0.1), rc[:, 7:8]], 1)
        mb = m.bool()
        compact = level >= 1                                         # eval.py:159
        r2 = recurse(sec[mb] if compact else sec, level + 1)
        part = r2["rgb_fine"]
        if compact:
            part = torch.zeros_like(r["rgb_fine"])
            part[mb] = r2["rgb_fine"]
        mf = mb.float()[:, None]
        r["rgb_fine"] = mf * part + (1 - mf) * r["rgb_fine"]
        return r

    outs = {}
    for i in range(0, rays.shape[0], chunk):
        for k, v in recurse(rays[i:i + chunk], 0).items():
            if v.dim() <= 2 and (v.dim() == 1 or v.shape[1] <= 3):
                outs.setdefault(k, []).append(v)
    return {k: torch.cat(v, 0) for k, v in outs.items()}


@torch.no_grad()
def render_train_coarse(models, rays, gt_mask, n_samples, chunk):
    """BASELINE config 1 on the host cores: train.NeRFSystem.forward (train.py:102-348) with coarse-only sampling
    (N_importance = 0), a valid ground-truth mirror mask and only_trace_rays_in_mirrors -- render, reflect the masked rays off the
    composited predicted normal (train.py:192-252), render those, blend (train.py:263-296).  Forward values only (no autograd:
    the density-gradient normal of compute_normal=True is left out; it is not read when the predicted normal exists)."""
    w = models["coarse"]

    def render(rc):
        rays_o, rays_d, near, far = rc[:, 0:3], rc[:, 3:6], rc[:, 6:7], rc[:, 7:8]
        t = torch.linspace(0, 1, n_samples)
        z = (near * (1 - t) + far * t).expand(rc.shape[0], n_samples)
        res = {}
        _inference(res, w, "coarse", rays_o, rays_d, z, embed(rays_d, 4), False, chunk)
        return res

    out = []
    for i in range(0, rays.shape[0], chunk):
        rc, mb = rays[i:i + chunk], gt_mask[i:i + chunk].bool()
        r = render(rc)
        rgb = r["rgb_coarse"]
        if bool(mb.any()):
            n = _l2n(r["surface_normal_coarse"])
            wv = _l2n(-rc[:, 3:6])
            rdir = 2 * (wv * n).sum(-1, keepdim=True) * n - wv
            sec = torch.cat([r["x_surface_coarse"], rdir, torch.full_like(rc[:, 6:7], 0.1), rc[:, 7:8]], 1)
            part = rgb.clone()
            part[mb] = render(sec[mb])["rgb_coarse"]
            mf = mb.float()[:, None]
            rgb = mf * part + (1 - mf) * rgb
        out.append(rgb)
    return torch.cat(out, 0)
