"""Plain-torch fp32 restatements of the two floating-point kernels that have a backward pass
(field MLP, compositing).  TEST INFRASTRUCTURE ONLY: the GPU tests differentiate these with
torch.autograd to check the hand-written HIP backward kernels; the product never imports this."""
import torch


def l2n(x):
    eps = torch.tensor(torch.finfo(torch.float32).eps, device=x.device, dtype=x.dtype)
    return x / torch.sqrt(torch.maximum((x * x).sum(-1, keepdim=True), eps))


def embed(x, n):
    out = [x]
    for k in range(n):
        out += [torch.sin((2.0 ** k) * x), torch.cos((2.0 ** k) * x)]
    return torch.cat(out, -1)


def field(w, xyz, dir_emb, with_normal=False, cut_normal=False, cut_mirror=False, keep_mirror=None):
    """w: dict name -> tensor (reference parameter names); returns sigma (B), rgb, pred_normal, is_mirror (B)
    [, normal = l2n(-d sigma/d xyz) built with create_graph=True like utils/func.py:10-25].
    cut_normal / cut_mirror / keep_mirror (B, bool): the heads see geo_feat.detach() (mirror_nerf.py:154-183)."""
    if with_normal and not xyz.requires_grad:
        xyz = xyz.requires_grad_(True)
    enc = embed(xyz, 10)
    h = enc
    for i in range(8):
        if i == 4:
            h = torch.cat([enc, h], -1)
        h = torch.relu(h @ w[f"xyz_encoding_{i+1}.0.weight"].T + w[f"xyz_encoding_{i+1}.0.bias"])
    sigma = (h @ w["sigma.weight"].T + w["sigma.bias"])[:, 0]
    fin = h @ w["xyz_encoding_final.weight"].T + w["xyz_encoding_final.bias"]
    hd = torch.relu(torch.cat([fin, dir_emb], -1) @ w["dir_encoding.0.weight"].T + w["dir_encoding.0.bias"])
    rgb = torch.sigmoid(hd @ w["rgb.0.weight"].T + w["rgb.0.bias"])
    hN = h.detach() if cut_normal else h
    hM = h.detach() if cut_mirror else h
    if keep_mirror is not None and not cut_mirror:
        hM = torch.where(keep_mirror[:, None], h, h.detach())
    hn = hN @ w["normal_net.0.weight"].T + w["normal_net.0.bias"]
    pn = l2n(hn @ w["normal_net.1.weight"].T + w["normal_net.1.bias"])
    hm = torch.nn.functional.leaky_relu(hM @ w["is_mirror_net.0.weight"].T + w["is_mirror_net.0.bias"], 0.01)
    m = torch.sigmoid(hm @ w["is_mirror_net.2.weight"].T + w["is_mirror_net.2.bias"])[:, 0]
    if with_normal:
        (grad,) = torch.autograd.grad(sigma, xyz, torch.ones_like(sigma), create_graph=True, retain_graph=True)
        return sigma, rgb, pn, m, l2n(-grad)
    return sigma, rgb, pn, m


def composite(rays, sigma, z, noise, rgb, is_mirror, pn, nrm, white_back=False, detach_mask=False, keep_mirror=None,
              detach_normal=False):
    """models/rendering.py:181-264, 362-367 on (N,S) tensors (detach_*: rendering.py:223-247)."""
    deltas = torch.cat([z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], 1e10)], -1)
    sv = sigma if noise is None else sigma + noise
    alphas = 1 - torch.exp(-deltas * torch.relu(sv))
    shifted = torch.cat([torch.ones_like(alphas[:, :1]), 1 - alphas + 1e-10], -1)
    w = alphas * torch.cumprod(shifted[:, :-1], -1)
    op = w.sum(1)
    out = {"weights": w, "opacity": op}
    rgb_map = (w[..., None] * rgb).sum(1)
    if white_back:
        rgb_map = rgb_map + 1 - op[:, None]
    depth = (w * z).sum(1)
    wm = w.detach() if detach_mask else w
    if keep_mirror is not None and not detach_mask:
        wm = torch.where(keep_mirror[:, None], w, w.detach())
    wn = w.detach() if detach_normal else w
    out.update(rgb=rgb_map, depth=depth, mask=(wm * is_mirror).sum(1), sn=(wn[..., None] * pn).sum(1))
    if nrm is not None:
        out["sng"] = (wn[..., None] * nrm).sum(1)
        out["nd"] = (wn * ((nrm - pn) ** 2).sum(-1)).sum(1)
    out["xs"] = rays[:, :3] + rays[:, 3:6] * depth[:, None]
    return out


def reflect(rays, x_surface, normal, mask, compact):
    """train.py:217-252 with torch ops (reference for ReflectFn)."""
    n = l2n(normal)
    w = l2n(-rays[:, 3:6])
    cos = (w * n).sum(-1)
    rdir = 2 * cos[:, None] * n - w
    far = rays[:, 7:8]
    sec = torch.cat([x_surface, rdir, torch.ones_like(far) * 0.1, far], -1)
    if compact:
        sec = sec[mask != 0]
    return sec


def blend(base, sec, mask, compact):
    """train.py:263-296 with torch ops (reference for BlendFn)."""
    if compact:
        part = base.clone().detach()
        part[mask != 0] = sec
    else:
        part = sec
    m = mask[:, None]
    return m * part + (1 - m) * base


def hashgrid_encode(x01, table, cfg):
    """Multiresolution hash encoding with torch ops (differentiable in `table` and, through the interpolation weights, in
    `x01`; twice differentiable): cells, hashing and level geometry follow the oracle (`hashgrid_encode` / `_grid_index`,
    i.e. models/gridencoder/src/gridencoder.cu:51-272), the table look-up is a gather.  x01 (B,3) fp32 in [0,1];
    table (n_entries, 2).  Returns (B, 2 * n_levels), level-major."""
    import numpy as np
    from oracle import mirror_nerf_oracle as O
    oob = ((x01 < 0) | (x01 > 1)).any(-1)
    feats = []
    for lv in range(cfg["n_levels"]):
        off0, off1 = int(cfg["offsets"][lv]), int(cfg["offsets"][lv + 1])
        scale = float(np.float32(np.exp2(np.float64(lv) * np.float64(cfg["S"])) * np.float64(cfg["H"]) - 1.0))
        res = int(np.ceil(np.float32(scale))) + 1
        pos = x01 * scale + 0.5
        pg = torch.floor(pos).detach()
        fr = (pos - pg).to(table.dtype)
        pgi = pg.cpu().numpy().astype(np.int64).clip(0).astype(np.uint32)
        acc = 0
        for c in range(8):
            wgt = 1
            loc = np.empty_like(pgi)
            for a in range(3):
                bit = (c >> a) & 1
                wgt = wgt * (fr[:, a] if bit else 1 - fr[:, a])
                loc[:, a] = pgi[:, a] + np.uint32(bit)
            idx = torch.from_numpy(O._grid_index(loc, off1 - off0, res)).to(x01.device) + off0
            acc = acc + wgt[:, None] * table[idx]
        feats.append(torch.where(oob[:, None], torch.zeros_like(acc), acc))
    return torch.cat(feats, -1)


def sh4(d):
    """Real spherical harmonics of degree 4 (16 values) of the raw direction: models/shencoder/src/shencoder.cu:49-79."""
    X, Y, Z = d[:, 0], d[:, 1], d[:, 2]
    xy, xz, yz, x2, y2, z2 = X * Y, X * Z, Y * Z, X * X, Y * Y, Z * Z
    return torch.stack([
        torch.full_like(X, 0.28209479177387814), -0.48860251190291987 * Y, 0.48860251190291987 * Z, -0.48860251190291987 * X,
        1.0925484305920792 * xy, -1.0925484305920792 * yz, 0.94617469575755997 * z2 - 0.31539156525251999,
        -1.0925484305920792 * xz, 0.54627421529603959 * x2 - 0.54627421529603959 * y2,
        0.59004358992664352 * Y * (-3.0 * x2 + y2), 2.8906114426405538 * xy * Z, 0.45704579946446572 * Y * (1.0 - 5.0 * z2),
        0.3731763325901154 * Z * (5.0 * z2 - 3.0), 0.45704579946446572 * X * (1.0 - 5.0 * z2),
        1.4453057213202769 * Z * (x2 - y2), 0.59004358992664352 * X * (-x2 + 3.0 * y2)], -1)


def tcnn_field_with_normal(w, x6, cfg):
    """tcnn_field plus normal = l2n(-d sigma/dx) built with create_graph=True (models/mirror_nerf_tcnn.py:172-218,
    utils/func.py:10-25): torch's double backward through it is the reference for the second-order kernel."""
    if not x6.requires_grad:
        x6 = x6.requires_grad_(True)
    sigma, rgb, pn, m = tcnn_field(w, x6, cfg)
    (grad,) = torch.autograd.grad(sigma, x6, torch.ones_like(sigma), create_graph=True, retain_graph=True)
    return sigma, rgb, pn, m, l2n(-grad[:, :3])


def tcnn_field(w, x6, cfg, detach_normal=False, detach_mirror=None):
    """MirrorNeRFTcnn.forward (models/mirror_nerf_tcnn.py:220-259) with torch ops: the hash-grid cells and interpolation
    weights follow the oracle (`hashgrid_encode` / `_grid_index`), the table look-up is a differentiable gather.
    w: dict of torch tensors (state_dict names), x6 (B,6) = [xyz, raw direction].  Returns sigma (B), rgb, pred_normal,
    is_mirror (B)."""
    xyz, d = x6[:, :3], x6[:, 3:6]
    bound = cfg["bound"]
    # cell coordinates in fp32 like the kernel (a cell of the finest level is 2e-4 of the box: fp64 coordinates would
    # differ from the kernel's by 1e-4 of a cell); everything after the interpolation weights runs in the dtype of w
    xf = xyz.float()
    x01 = (xf + bound) / torch.full_like(xf, 2 * bound)     # a tensor divisor: torch turns "/ scalar" into "* (1/scalar)" on the GPU
    enc = hashgrid_encode(x01, w["encoder.embeddings"], cfg)
    h = torch.relu(enc @ w["sigma_net.0.weight"].T) @ w["sigma_net.1.weight"].T
    sigma, geo = h[:, 0], h[:, 1:]
    # the --detach_density_* options (models/mirror_nerf_tcnn.py:186-215): detach_mirror = True (all samples) or a (B,) bool
    # tensor of the samples whose mirror head sees geo_feat.detach()
    geo_n = geo.detach() if detach_normal else geo
    pn = l2n(torch.relu(geo_n @ w["normal_net.0.weight"].T) @ w["normal_net.1.weight"].T)
    sh = sh4(d)
    hc = torch.relu(torch.cat([sh, geo], -1) @ w["color_net.0.weight"].T)
    hc = torch.relu(hc @ w["color_net.1.weight"].T)
    rgb = torch.sigmoid(hc @ w["color_net.2.weight"].T)
    if detach_mirror is None:
        geo_m = geo
    elif detach_mirror is True:
        geo_m = geo.detach()
    else:
        geo_m = torch.where(detach_mirror[:, None], geo.detach(), geo)
    hm = torch.nn.functional.leaky_relu(geo_m @ w["is_mirror_net.0.weight"].T + w["is_mirror_net.0.bias"], 0.01)
    m = torch.sigmoid(hm @ w["is_mirror_net.2.weight"].T + w["is_mirror_net.2.bias"])[:, 0]
    return sigma, rgb, pn, m
