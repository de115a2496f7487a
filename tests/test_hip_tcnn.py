"""Hash-grid field (BASELINE config 5).
Second half of this file (fixtures G17): the HIP path against the REFERENCE's own models/mirror_nerf_tcnn.py, models/rendering.py and
train.NeRFSystem(model_type="nerf_tcnn") run unchanged over stand-in encoders (tests/golden/make_golden_tcnn.py) -- everything
downstream of the encoder's interpolation is pinned.  The interpolation itself (tinycudann, not installable here) is UNPINNED: the
first half holds the kernels to this repository's restatement of gridencoder.cu / shencoder.cu (oracle, tests/torch_ref.py -- which
tests/test_oracle_golden.py in turn holds to the reference's G17 outputs and gradients)."""
import warnings

import numpy as np
import pytest
import torch

from oracle import mirror_nerf_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(bound, seed=0, table_scale=0.5):
    import mirror_nerf_amd as M
    torch.manual_seed(seed)
    m = M.MirrorNeRFTcnn(encoding="hashgrid", bound=bound, predict_normal=True, predict_mirror_mask=True)
    with torch.no_grad():   # the default 1e-4 table makes every output ~constant: use a livelier one
        m.encoder.embeddings.uniform_(-table_scale, table_scale)
    w = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
    return m.to(DEV), w, O.hashgrid_config(bound)


@pytest.mark.parametrize("bound,B", [(1.0, 700), (6.0, 1000)])
def test_tcnn_field_matches_oracle(bound, B):
    m, w, cfg = _model(bound)
    assert np.array_equal(cfg["offsets"], m.cfg["offsets"])
    rs = np.random.RandomState(B)
    xyz = rs.uniform(-bound, bound, (B, 3)).astype(np.float32)
    xyz[:5] *= 1.5                      # a few samples outside the box: encoding = 0 (gridencoder.cu:118-147)
    d = rs.normal(size=(B, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    x6 = np.concatenate([xyz, d], 1)
    want = O.tcnn_field_forward(w, x6, cfg, False, True)
    got = {k: v.detach().cpu().numpy() for k, v in m(torch.from_numpy(x6).to(DEV), compute_normal=True).items()}  # (autograd route)
    assert got["sigma"].shape == (B,) and got["is_mirror"].shape == (B, 1) and got["geo_feat"].shape == (B, 15)
    for k in ("sigma", "geo_feat", "rgb", "is_mirror"):
        err = float(np.max(np.abs(got[k] - want[k])))
        assert err <= 2e-5, (k, err)
    # l2-normalised 3-vectors amplify rounding by 1/|v|: tight in the median, bounded in the worst case
    dp = np.abs(got["pred_normal"] - want["pred_normal"]).max(-1)
    assert np.median(dp) <= 1e-4 and dp.max() <= 2e-2
    dn = np.abs(got["normal"] - want["normal"]).max(-1)
    assert np.median(dn) <= 1e-5 and np.mean(dn < 1e-3) >= 0.95
    with torch.no_grad():                                                                            # (inference route)
        so = {k: v.cpu().numpy() for k, v in m(torch.from_numpy(xyz).to(DEV), compute_normal=False, sigma_only=True).items()}
    assert "rgb" not in so and np.max(np.abs(so["sigma"] - want["sigma"])) <= 2e-5


def test_tcnn_render_rays_matches_oracle_compositing():
    """render_rays with the hash-grid models: per-sample field from the oracle restatement, compositing and
    resampling from the (pinned) oracle of rendering.py."""
    import mirror_nerf_amd as M
    # a +-0.5 random table at resolution 12288 changes by ~500 per unit length: 1e-5 of depth jitter between
    # two correct implementations would move outputs by 5e-3.  A gentler table keeps the comparison meaningful.
    mc, wc, cfg = _model(6.0, 1, table_scale=0.03)
    mf, wf, _ = _model(6.0, 2, table_scale=0.03)
    rays = O.synthetic_rays(12, 12)
    emb = {"xyz": M.Embedding(0), "dir": M.Embedding(0)}
    with torch.no_grad():
        got = M.render_rays({"coarse": mc, "fine": mf}, emb, torch.from_numpy(rays).to(DEV), 64, False, 0, 0, 64,
                            test_time=True, compute_normal=False)
    # oracle: same pipeline with the tcnn field plugged into the compositing of rendering.py
    N = rays.shape[0]
    z = O.render_rays.__globals__["torch_linspace"](0, 1, 64)
    zc = (rays[:, 6:7] * (1 - z) + rays[:, 7:8] * z).astype(np.float32)

    def field(w, zv, sigma_only):
        xyz = (rays[:, None, :3] + rays[:, None, 3:6] * zv[..., None]).reshape(-1, 3).astype(np.float32)
        x6 = np.concatenate([xyz, np.repeat(rays[:, 3:6], zv.shape[1], 0)], 1)
        return O.tcnn_field_forward(w, x6, cfg, sigma_only, False)

    def composite(sig, zv):
        deltas = np.concatenate([zv[:, 1:] - zv[:, :-1], np.full((N, 1), 1e10, np.float32)], 1)
        a = 1 - np.exp(-deltas * np.maximum(sig, 0))
        T = np.cumprod(np.concatenate([np.ones((N, 1)), 1 - a + 1e-10], 1)[:, :-1], 1)
        return (a * T).astype(np.float32)

    wco = composite(field(wc, zc, True)["sigma"].reshape(N, 64), zc)
    assert np.max(np.abs(got["weights_coarse"].cpu().numpy() - wco)) <= 1e-4
    mid = 0.5 * (zc[:, :-1] + zc[:, 1:])
    zf = np.sort(np.concatenate([zc, O.sample_pdf(mid, wco[:, 1:-1], 64, det=True)], 1), 1)
    o = field(wf, zf, False)
    wfi = composite(o["sigma"].reshape(N, 128), zf)
    rgb = (wfi[..., None] * o["rgb"].reshape(N, 128, 3)).sum(1)
    # self-consistency tolerance 5e-4: the fine depths of the two sides differ by ~1e-5 and this field is steep
    assert np.max(np.abs(got["rgb_fine"].cpu().numpy() - rgb)) <= 5e-4
    assert np.max(np.abs(got["mirror_mask_fine"].cpu().numpy() - (wfi * o["is_mirror"].reshape(N, 128)).sum(1))) <= 5e-4


# ----------------------------------------------------------------------------------------------------- training
def _grads_of(model, x6, seeds, which):
    """Gradients of sum_k <seed_k, out_k> from the HIP path: dict name -> tensor (+ 'x6')."""
    model.zero_grad()
    x = x6.clone().requires_grad_(True)
    out = model(x, compute_normal=False)
    outs = {"sigma": out["sigma"], "rgb": out["rgb"], "pred_normal": out["pred_normal"], "is_mirror": out["is_mirror"][:, 0]}
    loss = sum((outs[k] * seeds[k]).sum() for k in which)
    loss.backward()
    g = {k: (v.grad.clone() if v.grad is not None else torch.zeros_like(v)) for k, v in model.named_parameters()}
    g["x6"] = x.grad.clone()
    return g, outs


def _ref_grads(model, x6, seeds, which, cfg, double=False):
    from tests import torch_ref as R
    dt = torch.float64 if double else torch.float32
    w = {k: v.detach().to(dt).clone().requires_grad_(True) for k, v in model.state_dict().items()}
    x = x6.detach().to(dt).clone().requires_grad_(True)
    sigma, rgb, pn, m = R.tcnn_field(w, x, cfg)
    outs = {"sigma": sigma, "rgb": rgb, "pred_normal": pn, "is_mirror": m}
    loss = sum((outs[k] * seeds[k].to(dt)).sum() for k in which)
    loss.backward()
    g = {k: (v.grad if v.grad is not None else torch.zeros_like(v)).float() for k, v in w.items()}
    g["x6"] = x.grad.float()
    return g, outs


@pytest.mark.parametrize("bound,B,which", [
    (1.0, 700, ("sigma", "rgb", "pred_normal", "is_mirror")),     # ragged tile (700 = 2*256 + 188)
    (6.0, 1500, ("sigma", "rgb", "pred_normal", "is_mirror")),
    (1.0, 300, ("sigma",)), (1.0, 300, ("rgb",)), (1.0, 300, ("pred_normal",)), (1.0, 300, ("is_mirror",)),
])
def test_tcnn_backward_matches_torch_autograd(bound, B, which):
    """mnrf_tcnn_backward against torch.autograd through a plain-torch restatement of the field (tests/torch_ref.py;
    fp64 for the reference gradients, so the tolerance measures the kernel): table, MLP weights, positions, directions."""
    m, _w, cfg = _model(bound, seed=3, table_scale=0.3)
    g = torch.Generator().manual_seed(B)
    xyz = (torch.rand(B, 3, generator=g) * 2 - 1) * bound
    xyz[:4] *= 1.4                                         # outside the box: no encoding gradient
    d = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1)
    x6 = torch.cat([xyz, d], 1).to(DEV)
    seeds = {"sigma": torch.randn(B, generator=g).to(DEV), "rgb": torch.randn(B, 3, generator=g).to(DEV),
             "pred_normal": torch.randn(B, 3, generator=g).to(DEV), "is_mirror": torch.randn(B, generator=g).to(DEV)}
    got, outs = _grads_of(m, x6, seeds, which)
    want, routs = _ref_grads(m, x6, seeds, which, cfg, double=True)
    for k in ("sigma", "rgb", "is_mirror"):
        assert float((outs[k].detach() - routs[k].detach().float()).abs().max()) <= 2e-5, k
    for k, wv in want.items():
        gv = got[k]
        assert gv.shape == wv.shape, (k, gv.shape, wv.shape)
        scale = float(wv.abs().max()) + 1e-12
        err = float((gv - wv).abs().max())
        if k == "x6":
            # d/dx jumps across cell faces and the l2-normalised head amplifies: compare where well conditioned
            rel = ((gv - wv).abs().max(-1).values / (wv.abs().max(-1).values + 1e-3 * scale))
            assert float(rel.median()) <= 1e-4 and float((rel < 1e-2).float().mean()) > 0.97, (k, float(rel.median()))
        else:
            # table: the default scatter is fixed point, 2^-17 of a level's largest contribution per add (MNRF_TCNN_GRAD_FIXED)
            assert err <= (6e-5 if k == "encoder.embeddings" else 2e-5) * scale + 1e-7, (k, err, scale)
    # untouched table entries stay exactly zero
    assert int((got["encoder.embeddings"] != 0).sum()) <= B * 16 * 8 * 2


def test_tcnn_backward_packed_f16_table_gradient():
    """module.table_grad_f16 (flag MNRF_TCNN_GRAD_F16): the big hashed levels accumulate the table gradient in half2 with one
    packed atomic per entry, scaled by 2^10 -- tinycudann's gradient precision.  Against the default fp32 scatter on the same
    inputs: everything that is not the table is unchanged bit for bit; the table gradient agrees to f16 accuracy on the
    entries that matter (2e-3 of the largest entry absolute; 1 % relative on entries above 1 % of the largest), and the same
    entries are touched."""
    B = 6000
    m, _w, cfg = _model(6.0, seed=5, table_scale=0.3)
    g = torch.Generator().manual_seed(17)
    xyz = (torch.rand(B, 3, generator=g) * 2 - 1) * 6.0
    d = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1)
    x6 = torch.cat([xyz, d], 1).to(DEV)
    seeds = {"sigma": 1e-3 * torch.randn(B, generator=g).to(DEV), "rgb": 1e-3 * torch.randn(B, 3, generator=g).to(DEV),
             "pred_normal": 1e-3 * torch.randn(B, 3, generator=g).to(DEV), "is_mirror": 1e-3 * torch.randn(B, generator=g).to(DEV)}
    which = ("sigma", "rgb", "pred_normal", "is_mirror")
    m.table_grad_f16 = False
    m.table_grad_fixed = False          # the baseline of this comparison: two fp32 atomics per entry
    want, _ = _grads_of(m, x6, seeds, which)
    m.table_grad_f16 = True
    got, _ = _grads_of(m, x6, seeds, which)
    for k in want:
        if k == "encoder.embeddings":
            continue
        if k == "x6":
            assert torch.equal(got[k], want[k]), k
        else:      # the MLP gradients are sums of atomics over workgroups: order-dependent in the last bits
            assert float((got[k] - want[k]).abs().max()) <= 1e-5 * (float(want[k].abs().max()) + 1e-12), k
    a, b = got["encoder.embeddings"], want["encoder.embeddings"]
    from mirror_nerf_amd.mirror_nerf_tcnn import _offsets17
    offs = list(_offsets17(m.cfg))
    first16 = next(offs[l] for l in range(16) if offs[l + 1] - offs[l] > 262144)      # levels without private copies (mnrf_tcnn.hip plan_copies)
    # the copied (coarse) levels keep their fp32 route (atomics: equal up to the order of the adds)
    assert float((a[:first16] - b[:first16]).abs().max()) <= 1e-5 * float(b[:first16].abs().max())
    a, b = a[first16:], b[first16:]
    scale = float(b.abs().max())
    assert scale > 0 and int((b != 0).sum()) > 10000
    assert float((a - b).abs().max()) <= 2e-3 * scale
    big = b.abs() > 1e-2 * scale
    assert int(big.sum()) > 1000
    assert float(((a - b).abs()[big] / b.abs()[big]).max()) <= 1e-2
    assert float(((a != 0) != (b != 0)).float().mean()) < 1e-3      # (a contribution below f16's subnormal step)


@pytest.mark.parametrize("mode", ["normal", "mask", "outside"])
def test_tcnn_backward_detach_density_options(mode):
    """--detach_density_for_normal_loss / _for_mask_loss / _outside_mirror_for_mask_loss on the hash-grid field
    (models/mirror_nerf_tcnn.py:186-215): the head evaluated on geo_feat.detach() keeps its own weight gradients but sends
    nothing into sigma_net / the table / the positions.  Against torch autograd (fp64) with the same .detach()s; the option
    must move some gradient by an amount no tolerance could hide."""
    m, _w, cfg = _model(1.0, seed=5, table_scale=0.3)
    B = 600
    g = torch.Generator().manual_seed(17)
    xyz = (torch.rand(B, 3, generator=g) * 2 - 1)
    d = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1)
    x6 = torch.cat([xyz, d], 1).to(DEV)
    seeds = {"sigma": torch.randn(B, generator=g).to(DEV), "rgb": torch.randn(B, 3, generator=g).to(DEV),
             "pred_normal": torch.randn(B, 3, generator=g).to(DEV), "is_mirror": torch.randn(B, generator=g).to(DEV)}
    if mode != "normal":       # the random-init mirror head sends little into geo_feat: weigh its seed so that the cut shows
        seeds["is_mirror"] = seeds["is_mirror"] * 300.0
    gt_mask = (torch.rand(B, generator=g) < 0.4).float().to(DEV)
    kw = dict(detach_density_for_normal_loss=mode == "normal", detach_density_for_mask_loss=mode == "mask",
              detach_density_outside_mirror_for_mask_loss=mode == "outside", mirror_mask=gt_mask if mode == "outside" else None)

    def hip(**k):
        m.zero_grad()
        x = x6.clone().requires_grad_(True)
        out = m(x, compute_normal=False, **k)
        outs = {"sigma": out["sigma"], "rgb": out["rgb"], "pred_normal": out["pred_normal"], "is_mirror": out["is_mirror"][:, 0]}
        sum((outs[q] * seeds[q]).sum() for q in outs).backward()
        gr = {q: (v.grad.clone() if v.grad is not None else torch.zeros_like(v)) for q, v in m.named_parameters()}
        gr["x6"] = x.grad.clone()
        return gr
    got, plain = hip(**kw), hip()
    from tests import torch_ref as R
    w = {q: v.detach().double().clone().requires_grad_(True) for q, v in m.state_dict().items()}
    x = x6.detach().double().clone().requires_grad_(True)
    sigma, rgb, pn, mm = R.tcnn_field(w, x, cfg, detach_normal=mode == "normal",
                                      detach_mirror=True if mode == "mask" else ((gt_mask == 0) if mode == "outside" else None))
    outs = {"sigma": sigma, "rgb": rgb, "pred_normal": pn, "is_mirror": mm}
    sum((outs[q] * seeds[q].double()).sum() for q in outs).backward()
    moved = 0.0
    for q, wv in w.items():
        wv = (wv.grad if wv.grad is not None else torch.zeros_like(wv)).float()
        scale = float(wv.abs().max()) + 1e-12
        assert float((got[q] - wv).abs().max()) <= (6e-5 if q == "encoder.embeddings" else 2e-5) * scale + 1e-7, (mode, q)      # (table: fixed-point scatter)
        moved = max(moved, float((got[q] - plain[q]).abs().max()) / scale)
    assert moved > 0.05, (mode, moved)       # the option changed some gradient by more than 5 % of its tensor's largest entry
    # the cut head's own weights are unaffected
    head = "normal_net.1.weight" if mode == "normal" else "is_mirror_net.2.weight"
    assert float((got[head] - plain[head]).abs().max()) <= 1e-6 * float(plain[head].abs().max())      # (sums of atomics over workgroups)


def test_tcnn_training_step_through_render_rays():
    """render_rays with hash-grid models under autograd: gradients reach the table and every MLP of both models and
    agree with torch.autograd through the restated field + compositing on the same sample positions."""
    import mirror_nerf_amd as M
    from tests import torch_ref as R
    mc, _wc, cfg = _model(2.0, 5, table_scale=0.05)
    mf, _wf, _ = _model(2.0, 6, table_scale=0.05)
    rays = torch.from_numpy(O.synthetic_rays(6, 6)).to(DEV)
    rays[:, 6], rays[:, 7] = 2.5, 5.5
    N = rays.shape[0]
    emb = {"xyz": M.Embedding(0), "dir": M.Embedding(0)}
    target = torch.rand(N, 3, generator=torch.Generator().manual_seed(1)).to(DEV)
    res = M.render_rays({"coarse": mc, "fine": mf}, emb, rays, 16, False, 0, 0, 16, compute_normal=False)
    loss = ((res["rgb_coarse"] - target) ** 2).mean() + ((res["rgb_fine"] - target) ** 2).mean() \
        + 0.1 * res["mirror_mask_fine"].mean() + 0.01 * res["surface_normal_fine"].sum()
    loss.backward()
    for mdl, typ in ((mc, "coarse"), (mf, "fine")):
        z = res[f"z_vals_{typ}"].detach()
        S = z.shape[1]
        w = {k: v.detach().double().clone().requires_grad_(True) for k, v in mdl.state_dict().items()}
        xyz = (rays[:, None, :3] + rays[:, None, 3:6] * z[..., None]).reshape(-1, 3)
        x6 = torch.cat([xyz, rays[:, None, 3:6].expand(N, S, 3).reshape(-1, 3)], 1).double()
        sigma, rgb, pn, mir = R.tcnn_field(w, x6, cfg)
        comp = R.composite(rays.double(), sigma.view(N, S), z.double(), None, rgb.view(N, S, 3), mir.view(N, S),
                           pn.view(N, S, 3), None)
        ref = ((comp["rgb"] - target.double()) ** 2).mean()
        if typ == "fine":
            ref = ref + 0.1 * comp["mask"].mean() + 0.01 * comp["sn"].sum()
        ref.backward()
        for k, p in mdl.named_parameters():
            assert p.grad is not None, (typ, k)
            wv = (w[k].grad if w[k].grad is not None else torch.zeros_like(w[k])).float()   # (heads no loss reads)
            scale = float(wv.abs().max()) + 1e-12
            assert float((p.grad - wv).abs().max()) <= 1e-4 * scale + 1e-8, (typ, k, float((p.grad - wv).abs().max()), scale)


def test_tcnn_backward_run_aggregation_is_order_independent():
    """Densely sampled rays put long runs of consecutive samples into one cell of the coarse levels (summed inside the
    wave before the scatter, into per-XCD copies); the same samples in random order have no runs.  Table and weight
    gradients must agree, and so must the per-ray reduction of the position gradient."""
    from mirror_nerf_amd.mirror_nerf_tcnn import TcnnFieldFn
    m, _w, cfg = _model(1.0, seed=7, table_scale=0.2)
    g = torch.Generator().manual_seed(11)
    N, S = 37, 96                                                     # 3552 samples: ragged last tile
    o = (torch.rand(N, 3, generator=g) - 0.5) * 0.6
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
    rays = torch.cat([o, d, torch.zeros(N, 1), torch.ones(N, 1)], 1).to(DEV)
    z = (torch.linspace(0.0, 0.9, S)[None] + 0.004 * torch.rand(N, S, generator=g)).to(DEV)     # spacing 0.0095: runs up to level 5
    seeds = [torch.randn(N * S, generator=g).to(DEV), torch.randn(N * S, 3, generator=g).to(DEV),
             torch.randn(N * S, 3, generator=g).to(DEV), torch.randn(N * S, generator=g).to(DEV)]

    def grads(mode):
        m.zero_grad()
        if mode == "rays":
            outs = TcnnFieldFn.apply(m, S, None, rays, z, None, False, m.encoder.embeddings, *m.mlp_params())
            sd = seeds
        else:
            perm = torch.randperm(N * S, generator=g).to(DEV)
            xyz = (rays[:, None, :3] + rays[:, None, 3:6] * z[..., None]).reshape(-1, 3)
            x6 = torch.cat([xyz, rays[:, None, 3:6].expand(N, S, 3).reshape(-1, 3)], 1)[perm].contiguous()
            outs = TcnnFieldFn.apply(m, 1, x6, None, None, None, False, m.encoder.embeddings, *m.mlp_params())
            sd = [s[perm] for s in seeds]
        sum((o_ * s_).sum() for o_, s_ in zip(outs[:4], sd)).backward()
        return {k: v.grad.clone() for k, v in m.named_parameters()}

    a, b = grads("rays"), grads("shuffled")
    for k in a:
        scale = float(a[k].abs().max()) + 1e-12
        assert float((a[k] - b[k]).abs().max()) <= (6e-5 if k == "encoder.embeddings" else 2e-5) * scale, (k, float((a[k] - b[k]).abs().max()), scale)


def test_tcnn_train_recursion_matches_torch_field(monkeypatch):
    """NeRFSystem.forward (train semantics, GT mirror mask, compacted reflected rays, blend) with hash-grid models:
    gradients with the HIP field backward against the same pipeline with the field replaced by torch ops (autograd
    through tests/torch_ref.tcnn_field; every other node is the HIP autograd function in both runs).  Exercises
    dL/d position and dL/d direction: the reflected rays start at x_surface and point along the reflected normal."""
    from types import SimpleNamespace
    import mirror_nerf_amd as M
    from mirror_nerf_amd import mirror_nerf_tcnn as T
    from tests import torch_ref as R
    hp = SimpleNamespace(model_type="nerf_tcnn", bound=2.0, predict_normal=True, predict_mirror_mask=True, N_samples=24,
                         N_importance=24, use_disp=False, perturb=0, noise_std=0, chunk=4096, only_one_field=False,
                         trace_secondary_rays=True, max_recursive_level=1, only_trace_rays_in_mirrors=True, for_vis=False)
    torch.manual_seed(4)
    system = M.NeRFSystem(hp)
    with torch.no_grad():
        for mdl in (system.nerf_coarse, system.nerf_fine):
            mdl.encoder.embeddings.uniform_(-0.05, 0.05)
            # a random table is white noise at the fine levels: d(output)/d(position) ~ 1e4 and a reflected ray that moves
            # by one ulp changes its gradient by O(1) -- keep the five coarsest levels only, so that the comparison of two
            # fp32 implementations is well conditioned
            mdl.encoder.embeddings[int(mdl.cfg["offsets"][5]):] = 0
            # x10: opaque enough for surfaces (and reflected rays) to matter.  (Measured while writing this test: at x40
            # a 1e-7 relative change of the ray origins moves these gradients by 3-7 % in BOTH implementations and they
            # differ by as much; at x10 the two agree to 7e-5.)
            mdl.sigma_net[1].weight[0] *= 10.0
            # ... which also means that the comparison needs the forward on the arithmetic of the torch field: the fp32 VALU
            # kernel (the matrix-pipe kernel differs from it by ~1e-6 in sigma, an order of magnitude more than the 1e-7 above)
            mdl.mlp_on_valu = True
    system.to(DEV)
    rays = torch.from_numpy(O.synthetic_rays(8, 8)).to(DEV)
    rays[:, 6], rays[:, 7] = 2.5, 5.5
    N = rays.shape[0]
    gt = (torch.arange(N, device=DEV) % 3 != 0).float()
    target = torch.rand(N, 3, generator=torch.Generator().manual_seed(2)).to(DEV)

    def run():
        system.zero_grad()
        res = system(rays, {"mirror_mask": gt, "is_eval": False, "train_geometry_stage": False})
        loss = ((res["rgb_fine"] - target) ** 2).mean() + ((res["rgb_coarse"] - target) ** 2).mean() \
            + 0.05 * ((res["mirror_mask_fine"] - gt) ** 2).mean()
        loss.backward()
        return float(loss.detach()), {k: v.grad.clone() for k, v in system.named_parameters()}

    loss_hip, g_hip = run()

    class TorchField:
        @staticmethod
        def apply(module, spr, xyz6, rays_, z, dirs, want_normal, table, *params):
            n = rays_.shape[0]
            xyz = (rays_[:, None, :3] + rays_[:, None, 3:6] * z[..., None]).reshape(-1, 3)
            dd = (dirs if dirs is not None else rays_[:, 3:6])[:, None, :3].expand(n, spr, 3).reshape(-1, 3)
            w = dict(module.named_parameters())
            sigma, rgb, pn, mir = R.tcnn_field(w, torch.cat([xyz, dd], 1), module.cfg)
            # the density-gradient normal is a constant of the graph in both runs: take it from the kernel
            nrm = module.field(n * spr, rays=rays_.detach().contiguous(), z_vals=z.contiguous(), spr=spr,
                               dirs=None if dirs is None else dirs.detach().contiguous(), grad_normal=True)["normal"] \
                if want_normal else torch.empty(0, 3, device=rays_.device)
            return sigma, rgb, pn, mir, nrm, None

    monkeypatch.setattr(T, "TcnnFieldFn", TorchField)
    loss_ref, g_ref = run()
    assert abs(loss_hip - loss_ref) <= 1e-4 * max(1.0, abs(loss_ref))     # (fp32 torch GEMMs vs the kernel's fma chains, sigma x40)
    touched, bad = 0, []
    for k, wv in g_ref.items():
        scale = float(wv.abs().max())
        if scale == 0:
            assert float(g_hip[k].abs().max()) == 0, k
            continue
        touched += 1
        err = float((g_hip[k] - wv).abs().max())
        print(f"{k:45s} err {err:.3e} scale {scale:.3e}")
        if err > 2e-3 * scale:
            bad.append((k, err, scale))
    assert not bad, bad
    assert touched == 18       # every tensor but the coarse normal and mirror heads (no loss reads them)


def test_tcnn_forward_sees_updated_weights():
    """The kernel reads the MLP weights through the scalar cache (constant address space): a changed blob at the SAME
    address (optimiser step, in-place edit) must be seen by the next launch."""
    m, w, cfg = _model(1.0, seed=9)
    rs = np.random.RandomState(5)
    x6 = np.concatenate([rs.uniform(-1, 1, (513, 3)), rs.normal(size=(513, 3))], 1).astype(np.float32)
    xt = torch.from_numpy(x6).to(DEV)
    with torch.no_grad():
        first = m(xt, compute_normal=False)["rgb"].cpu().numpy()
        ptrs, rounds = {m._weights().data_ptr()}, 0
        # (the caching allocator hands the blob's address out again after a few rounds -- how many depends on what the
        # process allocated before: at least 4 rounds, then until an address has repeated)
        while rounds < 4 or (len(ptrs) == rounds + 1 and rounds < 64):
            rounds += 1
            for name in ("sigma_net.0.weight", "color_net.1.weight", "is_mirror_net.0.bias"):
                p = dict(m.named_parameters())[name]
                p.mul_(1.25 if rounds % 2 else 0.8).add_(0.01 if rounds % 2 else -0.008)      # (there and back: bounded)
                w[name] = p.detach().cpu().numpy().copy()
            ptrs.add(m._weights().data_ptr())     # (the caching allocator alternates between two blocks: addresses repeat)
            got = {k: v.cpu().numpy() for k, v in m(xt, compute_normal=False).items()}
            want = O.tcnn_field_forward(w, x6, cfg, False, False)
            for k in ("sigma", "rgb", "is_mirror"):
                assert float(np.max(np.abs(got[k] - want[k]))) <= 5e-5, k
            assert float(np.max(np.abs(first - got["rgb"]))) > 1e-3      # (the edit did change the output)
            first = got["rgb"]
    assert len(ptrs) <= rounds                                    # (at least one address was reused)


@pytest.mark.parametrize("sigma_only,grad", [(False, False)])
def test_tcnn_mfma_kernel_agrees_with_valu_kernel(sigma_only, grad):
    """The full evaluation without the density-gradient normal runs with its MLPs as hi/lo f16 tiles on the matrix pipe, four
    lanes per sample sharing its 16 levels; here against the fp32 VALU kernel (one thread per sample; it keeps the
    sigma-only and density-gradient launches) on a table with every level populated: the same cells and interpolation
    weights, ~2^-20 relative per product in the MLPs."""
    import mirror_nerf_amd as M
    torch.manual_seed(7)
    m = M.MirrorNeRFTcnn(encoding="hashgrid", bound=3.0, predict_normal=True, predict_mirror_mask=True).to(DEV)
    with torch.no_grad():
        m.encoder.embeddings.uniform_(-0.5, 0.5)
        m.sigma_net[1].weight[0] *= 5.0
    B = 3000                                              # ragged: 5 full workgroup tiles of 512 + 440
    x6 = torch.cat([(torch.rand(B, 3, device=DEV) * 2 - 1) * 3.2, torch.nn.functional.normalize(torch.randn(B, 3, device=DEV), dim=1)], 1)
    x6[::97, 0] = 3.5                                     # some samples outside the box (zero features)
    outs = []
    for valu in (True, False):
        m.mlp_on_valu = valu
        outs.append(m.field(B, xyz=x6.contiguous(), xyz_stride=6, sigma_only=sigma_only, grad_normal=grad, want_geo=True))
    a, b = outs
    assert set(a) == set(b)
    for k in a:
        d = (a[k] - b[k]).abs()
        if k == "normal":       # normalised gradient: compare where it is well conditioned
            assert float(d.max(-1)[0].median()) <= 1e-5 and float((d.max(-1)[0] < 1e-3).float().mean()) > 0.95
        else:
            assert float(d.max()) <= 2e-5 * max(1.0, float(a[k].abs().max())), (k, float(d.max()))


@pytest.mark.parametrize("bound,B", [(1.0, 700), (3.0, 1200)])
def test_tcnn_second_order_backward_matches_double_backward(bound, B):
    """The gradient that reaches the hash table, sigma_net and the positions through normal = l2n(-d sigma/dx)
    (models/mirror_nerf_tcnn.py:172-218 differentiates autograd.grad(sigma, x, create_graph=True)): tcnn_bwd2_kernel against
    torch's double backward through the plain-torch restatement, in fp64."""
    from tests import torch_ref as R
    m, _w, cfg = _model(bound, seed=5, table_scale=0.3)
    g = torch.Generator().manual_seed(B)
    xyz = (torch.rand(B, 3, generator=g) * 2 - 1) * bound * 0.98
    xyz[:3] *= 1.5                                         # outside the box
    d = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1)
    x6 = torch.cat([xyz, d], 1).to(DEV)
    cot_n = torch.randn(B, 3, generator=g).to(DEV)
    cot_s = torch.randn(B, generator=g).to(DEV) * 0.1
    # HIP
    m.zero_grad()
    x = x6.clone().requires_grad_(True)
    out = m(x, compute_normal=True)
    ((out["normal"] * cot_n).sum() + (out["sigma"] * cot_s).sum()).backward()
    got = {k: (v.grad.clone() if v.grad is not None else torch.zeros_like(v)) for k, v in m.named_parameters()}
    got["x6"] = x.grad.clone()
    # torch, fp64
    w = {k: v.detach().double().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    xr = x6.detach().double().clone().requires_grad_(True)
    sigma, rgb, pn, mm, nrm = R.tcnn_field_with_normal(w, xr, cfg)
    ((nrm * cot_n.double()).sum() + (sigma * cot_s.double()).sum()).backward()
    # the normals themselves agree where the gradient is well conditioned
    dn = (out["normal"].detach() - nrm.detach().float()).abs().max(-1).values
    assert float(dn.median()) <= 1e-5
    for k in ("encoder.embeddings", "sigma_net.0.weight", "sigma_net.1.weight"):
        wv = w[k].grad.float()
        scale = float(wv.abs().max()) + 1e-12
        err = float((got[k] - wv).abs().max())
        print(f"{k:24s} err {err:.3e} scale {scale:.3e}")
        assert err <= 2e-3 * scale, (k, err, scale)
    for k in ("color_net.0.weight", "normal_net.0.weight", "is_mirror_net.0.weight"):
        assert float(got[k].abs().max()) == 0.0            # untouched by this loss
    gx, wx = got["x6"][:, :3], xr.grad.float()[:, :3]
    rel = (gx - wx).abs().max(-1).values / (wx.abs().max(-1).values + 1e-3 * float(wx.abs().max()))
    assert float(rel.median()) <= 1e-3 and float((rel < 5e-2).float().mean()) > 0.9, (float(rel.median()), float((rel < 5e-2).float().mean()))


# ============================================================================================ fixtures G17
# The reference's own models/mirror_nerf_tcnn.py (+ models/rendering.py, train.NeRFSystem, torch.autograd) run unchanged over
# stand-in encoders (tests/golden/make_golden_tcnn.py): what the kernels are compared with below is the REFERENCE's output,
# not this repository's restatement.  Encoder interpolation unpinned (tinycudann absent), everything downstream pinned.
def _g17_model(fx, prefix, which=0):
    import mirror_nerf_amd as M
    from tests.golden import fixtures as FX
    w = FX.tcnn_weights(fx, prefix, which)
    cfg = w.pop("_cfg")
    m = M.MirrorNeRFTcnn(encoding="hashgrid", bound=fx.meta["table"]["bound"], cuda_ray=False, density_scale=1, min_near=0.2,
                         density_thresh=10, bg_radius=False, predict_normal=True, predict_mirror_mask=True)   # train.py:71-82
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    return m.to(DEV), cfg


@pytest.mark.parametrize("name", ["g17_tcnn_field_b1", "g17_tcnn_field_b6"])
def test_g17_field_forward(name):
    """MirrorNeRFTcnn.forward (models/mirror_nerf_tcnn.py:151-259) full + autograd normal, and sigma_only."""
    from tests.golden import fixtures as FX
    fx = FX.Fixture(name)
    m, _cfg = _g17_model(fx, "w__")
    x6 = torch.from_numpy(fx.inputs["x6"]).to(DEV)
    B = x6.shape[0]
    full = {k: v.detach().cpu().numpy() for k, v in m(x6.clone(), compute_normal=True).items()}       # (autograd route)
    with torch.no_grad():
        full_ng = {k: v.cpu().numpy() for k, v in m(x6.clone(), compute_normal=True).items()}           # (inference route)
        so = {k: v.cpu().numpy() for k, v in m(x6[:, :3].contiguous(), compute_normal=False, sigma_only=True).items()}
    assert set(so) == {"sigma", "geo_feat", "pred_normal"} and full["sigma"].shape == (B,) and full["is_mirror"].shape == (B, 1)
    want = {k.split("__")[1]: v for k, v in fx.outputs.items() if k.startswith("full__")}
    assert list(full) == list(want)           # the reference's key order too (normal first, mirror_nerf_tcnn.py:172-185)
    for got in (full, full_ng):
        for k in ("sigma", "geo_feat", "rgb", "is_mirror"):
            err = float(np.max(np.abs(got[k] - want[k])))
            assert err <= 2e-5, (k, err)
        # l2-normalised 3-vectors amplify rounding by 1/|v| (the reference's own fp32-vs-fp64 floor is in the fixture)
        dp = np.abs(got["pred_normal"] - want["pred_normal"]).max(-1)
        assert np.median(dp) <= 1e-4 and dp.max() <= max(2e-2, 4 * fx.meta["floor"]["pred_normal"])
        dn = np.abs(got["normal"] - want["normal"]).max(-1)
        assert np.median(dn) <= 1e-5 and np.mean(dn < 1e-3) >= 0.95
    assert np.max(np.abs(so["sigma"] - fx.outputs["sigma_only__sigma"])) <= 2e-5
    assert np.max(np.abs(so["geo_feat"] - fx.outputs["sigma_only__geo_feat"])) <= 2e-5


@pytest.mark.parametrize("variant", ["plain", "detach_normal", "detach_mask", "detach_outside", "second_order"])
def test_g17_field_gradients(variant):
    """mnrf_tcnn_backward against the REFERENCE's autograd through mirror_nerf_tcnn.py: every MLP tensor, the table (per-level
    norms + 4096 sampled entries + the number of touched values), positions and directions -- with no flag, with each of the
    three --detach_density_* options (186-215), and the second-order term through normal = l2n(-d sigma/dx) (172-180)."""
    from tests.golden import fixtures as FX
    fx = FX.Fixture("g17_tcnn_field_grads")
    m, cfg = _g17_model(fx, "w__")
    o = fx.outputs
    x = torch.from_numpy(fx.inputs["x6"]).to(DEV).requires_grad_(True)
    cot = {k[5:]: torch.from_numpy(v).to(DEV) for k, v in fx.inputs.items() if k.startswith("cot__")}
    second = variant == "second_order"
    kw = {"detach_normal": dict(detach_density_for_normal_loss=True), "detach_mask": dict(detach_density_for_mask_loss=True),
          "detach_outside": dict(detach_density_outside_mirror_for_mask_loss=True,
                                 mirror_mask=torch.from_numpy(fx.inputs["inside"]).to(DEV))}.get(variant, {})
    out = m(x, compute_normal=second, **kw)
    keys = ("sigma", "normal") if second else ("sigma", "rgb", "pred_normal", "is_mirror")
    loss = sum((out[k] * cot[k]).sum() for k in keys)
    loss.backward()
    want_loss = float(o[f"{variant}__loss"])
    assert abs(loss.item() - want_loss) <= (2e-3 if second else 2e-5) * abs(want_loss)
    tol = 2e-3 if second else 1e-4
    for k, p in m.named_parameters():
        if k == "encoder.embeddings":
            continue
        want = o[f"{variant}__grad__{k}"]
        g = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(want)
        err, scale = float(np.max(np.abs(g - want))), float(np.abs(want).max())
        assert err <= tol * scale + 1e-8, (variant, k, err, scale)
    lv, val, nnz = FX.table_grad_summary(m.encoder.embeddings.grad.cpu().numpy(), cfg, o[f"{variant}__table_idx"])
    wl, wv = o[f"{variant}__table_levels"], o[f"{variant}__table_val"]
    assert abs(nnz - int(o[f"{variant}__table_nnz"])) <= 0.015 * int(o[f"{variant}__table_nnz"])    # (contributions below the fixed-point step of the default scatter)
    assert np.max(np.abs(lv[:, 1] - wl[:, 1])) <= tol * wl[:, 1].max(), (lv[:, 1], wl[:, 1])
    assert np.max(np.abs(val - wv)) <= tol * np.abs(wv).max() * (5 if second else 1)
    gx, wx = x.grad.cpu().numpy(), o[f"{variant}__grad__x6"]
    # d/dx jumps across cell faces and the l2-normalised heads amplify: compare where well conditioned
    rel = np.abs(gx - wx).max(-1) / (np.abs(wx).max(-1) + 1e-3 * np.abs(wx).max())
    assert np.median(rel) <= (1e-3 if second else 1e-4) and np.mean(rel < (5e-2 if second else 1e-2)) > (0.9 if second else 0.97)


@pytest.mark.parametrize("name", ["g17_tcnn_render_train", "g17_tcnn_render_test"])
def test_g17_render_rays(name):
    """models/rendering.py:render_rays (64 + 64) with a coarse and a fine hash-grid model, as captured from the reference."""
    import mirror_nerf_amd as M
    from tests.golden import fixtures as FX
    fx = FX.Fixture(name)
    mc, _ = _g17_model(fx, "coarse__", 0)
    mf, _ = _g17_model(fx, "fine__", 1)
    tt = fx.meta["test_time"]
    emb = {"xyz": M.Embedding(0), "dir": M.Embedding(0)}
    with torch.no_grad():
        got = M.render_rays({"coarse": mc, "fine": mf}, emb, torch.from_numpy(fx.inputs["rays"]).to(DEV), 64, False, 0, 0, 64,
                            32768, False, tt, compute_normal=not tt)
    n = 0
    for k, want in fx.outputs.items():
        if k in FX.PER_SAMPLE_FINE:
            continue
        assert k in got, k
        err = float(np.max(np.abs(got[k].cpu().numpy().astype(np.float64) - want))) if want.size else 0.0
        assert err <= FX.tolerance(k, fx.meta), (k, err, FX.tolerance(k, fx.meta))
        n += 1
    assert n >= (8 if tt else 18)
    assert set(got) >= set(fx.outputs)


@pytest.mark.parametrize("pinned", [False, True])
@pytest.mark.parametrize("name", ["g17_tcnn_train_grads", "g17_tcnn_train_grads_full"])
def test_g17_train_step_gradients(name, pinned, monkeypatch):
    """train.NeRFSystem(model_type="nerf_tcnn").forward (train.py:67-99, 102-348: GT mirror mask, compacted reflected rays
    starting at x_surface along the reflected normal, blend) + loss + the reference autograd's gradient of every parameter
    of both models incl. the tables; `_full` adds the terms on the autograd normal (second order).
    The models hold a SMOOTH table (tests/golden/weights.make_smooth_tcnn_table) and the ray set was screened on the reference
    (make_golden_tcnn.train_case): with a white-noise table the reference's own gradients move by tens of percent under 1e-6
    perturbations of the rays.  Tolerance per tensor: 1e-3 of its largest entry, or 4 x what the reference's gradient moves in
    float64 / under 1e-6 ray perturbations (stored per tensor).  `pinned`: the reference's fine depths of both recursion
    levels are injected (render_rays(_z_fine=...)), which removes the inverse-CDF's sensitivity from the comparison."""
    from types import SimpleNamespace
    import mirror_nerf_amd as M
    from mirror_nerf_amd import recursion as RC
    from tests.golden import fixtures as FX
    from tests.golden import make_golden_loss as GL
    fx = FX.Fixture(name)
    loss_fn = getattr(GL, fx.meta["loss"])
    hp = dict(fx.meta["hp"])
    hp.update(model_type="nerf_tcnn", bound=fx.meta["table"]["bound"], predict_normal=True, predict_mirror_mask=True)
    system = M.NeRFSystem(SimpleNamespace(**hp))
    for mod, (prefix, which) in ((system.nerf_coarse, ("coarse__", 0)), (system.nerf_fine, ("fine__", 1))):
        w = FX.tcnn_weights(fx, prefix, which)
        cfg = w.pop("_cfg")
        mod.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    system.to(DEV)
    t = lambda k: torch.from_numpy(fx.inputs[k]).to(DEV)  # noqa: E731
    if pinned:
        orig, level = RC.render_rays, [0]

        def with_reference_depths(models, emb, rays, *a, **k):
            z = torch.from_numpy(fx.outputs[f"z_fine_l{level[0]}"]).to(DEV)
            level[0] += 1
            assert z.shape[0] == rays.shape[0]
            return orig(models, emb, rays, *a, **dict(k, _z_fine=z))
        monkeypatch.setattr(RC, "render_rays", with_reference_depths)
    res = system(t("rays"), {"mirror_mask": t("gt_mask"), "is_eval": False, "train_geometry_stage": False})
    n_cmp = 0
    for k, want in fx.outputs.items():
        if k == "loss" or k.startswith(("grad__", "table_", "z_fine_l")) or k in FX.PER_SAMPLE_FINE or k not in res:
            continue
        d = float(np.max(np.abs(res[k].detach().cpu().numpy().astype(np.float64) - want))) if want.size else 0.0
        assert d <= FX.tolerance(k, fx.meta), (k, d)
        n_cmp += 1
    assert n_cmp >= 20
    loss = loss_fn(res, t("target"), t("gt_mask"))
    assert abs(loss.item() - float(fx.outputs["loss"])) <= (2e-3 if fx.meta["loss"] == "full_loss" else 2e-5)
    loss.backward()
    floors = [fx.meta["grad_floors"], fx.meta["grad_pinned_floors" if pinned else "grad_free_floors"]]
    floor = lambda key: max(f.get(key, 0.0) for f in floors)  # noqa: E731
    report = []
    for mname, mod in (("coarse", system.nerf_coarse), ("fine", system.nerf_fine)):
        for pn_, p_ in mod.named_parameters():
            if pn_ == "encoder.embeddings":
                lv, val, _nnz = FX.table_grad_summary(p_.grad.cpu().numpy(), cfg, fx.outputs[f"table_idx__{mname}"])
                wl, wv = fx.outputs[f"table_levels__{mname}"], fx.outputs[f"table_val__{mname}"]
                fl = floor(f"{mname}__encoder.params")
                report.append((np.max(np.abs(lv[:, 1] - wl[:, 1])) / wl[:, 1].max(), max(1e-3, 4 * fl), mname, "table level norms"))
                report.append((np.max(np.abs(val - wv)) / np.abs(wv).max(), max(1e-3, 4 * fl), mname, "table entries"))
                continue
            key = f"grad__{mname}__{pn_}"
            want = fx.outputs[key]
            g = p_.grad.cpu().numpy() if p_.grad is not None else np.zeros_like(want)
            if np.abs(want).max() == 0:
                assert np.abs(g).max() == 0, key
                continue
            report.append((np.max(np.abs(g - want)) / np.abs(want).max(), max(1e-3, 4 * floor(f"{mname}__{pn_}")), mname, pn_))
    report.sort(key=lambda r: -r[0] / r[1])
    print("G17 relative gradient errors / tolerance, worst first:")
    for err, tol, mname, pn_ in report[:8]:
        print(f"  {err:.2e} / {tol:.1e} {mname} {pn_}")
    bad = [r for r in report if r[0] > r[1]]
    assert not bad, bad[:4]
    assert len(report) >= 22


def test_tcnn_single_pass_f16_mlp_and_level_major_encoding():
    """module.mlp_f16 (MNRF_TCNN_F16): the MLPs as single-pass f16 products on the matrix pipe -- "fp16 MLP on CDNA4 MFMA"
    (BASELINE config 5; tinycudann under precision=16, train.py:586).  Against the fp32-accurate default on the same inputs:
    f16 accuracy (a few 1e-3 of each output's scale), full AND sigma-only launches (the latter move to the matrix pipe).
    And mnrf_tcnn_encode (the level-major encoding launch) against the oracle's encoding; the two-launch forward against the
    one-launch form."""
    from mirror_nerf_amd import _lib
    from mirror_nerf_amd.mirror_nerf_tcnn import _offsets17
    m, w, cfg = _model(6.0, seed=2, table_scale=0.2)
    rays = torch.from_numpy(O.synthetic_rays(16, 16)).to(DEV)
    N, S = rays.shape[0], 48
    z = (torch.linspace(0.3, 7.5, S, device=DEV)[None] + 0.05 * torch.rand(N, S, device=DEV)).contiguous()
    with torch.no_grad():
        ref = m.field(N * S, rays=rays, z_vals=z, spr=S, want_geo=True)
        ref_s = m.field(N * S, rays=rays, z_vals=z, spr=S, sigma_only=True)
        m.mlp_f16 = True
        got = m.field(N * S, rays=rays, z_vals=z, spr=S, want_geo=True)
        got_s = m.field(N * S, rays=rays, z_vals=z, spr=S, sigma_only=True)
        m.mlp_f16 = False
    assert torch.equal(ref["sigma"], m.field(N * S, rays=rays, z_vals=z, spr=S)["sigma"])      # the flag is per call
    for k in ("sigma", "rgb", "is_mirror", "geo_feat"):
        scale = max(1.0, float(ref[k].abs().max()))
        err = float((got[k] - ref[k]).abs().max())
        assert 0 < err <= 4e-3 * scale, (k, err, scale)              # f16 operands: different, and close
    dp = (got["pred_normal"] - ref["pred_normal"]).abs().max(-1).values
    assert float(dp.median()) <= 2e-3
    assert float((got_s["sigma"] - ref_s["sigma"]).abs().max()) <= 4e-3 * max(1.0, float(ref_s["sigma"].abs().max()))
    assert torch.equal(got_s["sigma"], got["sigma"])                 # sigma-only on the matrix pipe: the same arithmetic as the full launch
    # the encoding on its own (mnrf_tcnn_encode: the level-major first launch of the two-launch forward) against the oracle's
    planes = torch.empty(16, N * S, 2, device=DEV)
    table = m.encoder.embeddings.detach().contiguous()
    _lib.check(_lib.lib().mnrf_tcnn_encode(_lib.ptr(table), _offsets17(m.cfg), m.cfg["S"], m.cfg["H"], float(m.bound), N * S, None, 0,
                                           _lib.ptr(rays), _lib.ptr(z), S, _lib.ptr(planes), _lib.stream()), "encode")
    xyz = (rays[:, None, :3] + rays[:, None, 3:6] * z[..., None]).reshape(-1, 3).cpu().numpy()
    enc = O.hashgrid_encode(((xyz + np.float32(6.0)) / np.float32(12.0)).astype(np.float32), w["encoder.embeddings"], cfg)
    got_enc = planes.permute(1, 0, 2).reshape(N * S, 32).cpu().numpy()
    assert np.max(np.abs(got_enc - enc)) <= 2e-6 * max(1.0, float(np.abs(enc).max()))
    # ... and the two-launch forward (default from 32768 samples on) equals the one-launch form bit for bit
    with torch.no_grad():
        m.enc_planes_min = 1
        two = m.field(N * S, rays=rays, z_vals=z, spr=S)
        m.enc_planes_min = 1 << 62
        one = m.field(N * S, rays=rays, z_vals=z, spr=S)
        del m.enc_planes_min
    for k in ("sigma", "rgb", "pred_normal", "is_mirror"):
        assert torch.equal(two[k], one[k]), k


def test_tcnn_packed_f16_gradient_overflow_is_clamped_and_reported():
    """MNRF_TCNN_GRAD_F16: a table-gradient sum beyond the f16 range (|g| * 2^10 > 65504) is clamped -- no inf / nan in
    d_table -- and reported one backward later, when the module falls back to fp32 atomics."""
    m, _w, cfg = _model(1.0, seed=4, table_scale=0.2)
    m.table_grad_f16 = True
    g = torch.Generator().manual_seed(3)
    B = 512
    x6 = torch.cat([torch.rand(B, 3, generator=g) * 0.02 + 0.4, torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1)], 1).to(DEV)

    def step(scale):
        m.zero_grad()
        out = m(x6.clone().requires_grad_(True), compute_normal=False)
        (out["sigma"].sum() * scale).backward()
        return m.encoder.embeddings.grad
    gtab = step(1e4)                                   # all samples in a few cells of the finest levels: sums far beyond 64
    off = cfg["offsets"]
    first16 = int(off[min(l for l in range(16) if off[l + 1] - off[l] > 262144)])     # levels without private fp32 copies: the half2 table
    assert bool(torch.isfinite(gtab).all()) and float(gtab[first16:].abs().max()) <= 65504.0 / 1024 * 1.001
    assert float(gtab[:first16].abs().max()) > 64.0                                      # (the fp32 levels show how large the sums are)
    assert m.table_grad_f16 is True
    with pytest.warns(RuntimeWarning, match="left the f16 range"):
        step(1.0)
    assert m.table_grad_f16 is False
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert bool(torch.isfinite(step(1e4)).all())   # fp32 atomics now


def test_tcnn_fixed_point_table_gradient():
    """MNRF_TCNN_GRAD_FIXED (default): one packed 64-bit integer atomic per table entry under a per-level power-of-two scale taken
    from the step's own largest |dL/d encoding|.  Against the fp32-atomic scatter on the same inputs: everything that is not the
    table is unchanged, the same entries are touched (but for contributions below the fixed-point step), the table gradient agrees to
    4e-5 of each level's largest entry;
    and the integer sums are ORDER-INDEPENDENT: two runs give bit-identical gradients (fp32 atomics do not)."""
    from mirror_nerf_amd.mirror_nerf_tcnn import TcnnFieldFn
    m, _w, cfg = _model(6.0, seed=8, table_scale=0.2)
    g = torch.Generator().manual_seed(21)
    N, S = 83, 96                                                     # 7968 samples: ragged last tile
    rays = torch.from_numpy(O.synthetic_rays(10, 10)[:N]).to(DEV)
    z = (torch.linspace(0.3, 7.9, S)[None] + 0.02 * torch.rand(N, S, generator=g)).to(DEV)   # through the box and out of it
    seeds = [torch.randn(N * S, generator=g).to(DEV), torch.randn(N * S, 3, generator=g).to(DEV),
             torch.randn(N * S, 3, generator=g).to(DEV), torch.randn(N * S, generator=g).to(DEV)]
    seeds[0][::7] = 0

    def grads(fixed):
        m.table_grad_fixed = fixed
        m.zero_grad()
        r = rays.clone().requires_grad_(True)
        outs = TcnnFieldFn.apply(m, S, None, r, z, None, False, m.encoder.embeddings, *m.mlp_params())
        sum((o_ * s_).sum() for o_, s_ in zip(outs[:4], seeds)).backward()
        out = {k: v.grad.clone() for k, v in m.named_parameters()}
        out["rays"] = r.grad.clone()
        return out
    a, a2, b = grads(True), grads(True), grads(False)
    del m.table_grad_fixed
    for k in a:
        if k != "encoder.embeddings":      # (the MLP gradients are sums of fp32 atomics over workgroups: order-dependent in the last bits)
            assert float((a[k] - b[k]).abs().max()) <= 1e-5 * (float(b[k].abs().max()) + 1e-12), k
    ta, tb = a["encoder.embeddings"], b["encoder.embeddings"]
    off = cfg["offsets"]
    first_fx = int(off[min(l for l in range(16) if off[l + 1] - off[l] > 262144)])     # (the coarse levels keep fp32 private copies)
    assert torch.equal(ta[first_fx:], a2["encoder.embeddings"][first_fx:])             # exact integer sums: reproducible bit for bit
    n_fixed = 0
    for lv in range(16):
        sa, sb = ta[off[lv]:off[lv + 1]], tb[off[lv]:off[lv + 1]]
        scale = float(sb.abs().max())
        if scale == 0:
            continue
        err = float((sa - sb).abs().max())
        assert err <= 4e-5 * scale, (lv, err, scale)                  # (step = S 2^-30, S = the level's sum of gradient magnitudes)
        if off[lv + 1] - off[lv] > 262144:
            n_fixed += 1
            assert int(((sa != 0) != (sb != 0)).sum()) <= 1e-2 * int((sb != 0).sum()) + 1      # (contributions below the fixed-point step)
    assert n_fixed >= 8


@pytest.mark.parametrize("name", ["g17_tcnn_eval_l1", "g17_tcnn_eval_l2"])
def test_g17_eval_recursion(name):
    """eval.batched_inference (eval.py:114-172, 293-360, 513-548, 676-740) with the hash-grid pair: predicted mirror mask thresholded
    in place, level 0 traces every ray of the chunk, deeper levels compact -- as captured from the reference."""
    import mirror_nerf_amd as M
    from tests.golden import fixtures as FX
    fx = FX.Fixture(name)
    m = fx.meta
    mc, _ = _g17_model(fx, "coarse__", 0)
    mf, _ = _g17_model(fx, "fine__", 1)
    emb = {"xyz": M.Embedding(0), "dir": M.Embedding(0)}
    got = M.batched_inference({"coarse": mc, "fine": mf}, emb, torch.from_numpy(fx.inputs["rays"]).to(DEV), m["N_samples"], m["N_importance"],
                              False, m["chunk"], args=m["args"], trace_secondary_rays=True, to_cpu=False)
    n = 0
    for k, want in fx.outputs.items():
        if k in FX.PER_SAMPLE_FINE:
            continue
        assert k in got, k
        err = float(np.max(np.abs(got[k].cpu().numpy().astype(np.float64) - want))) if want.size else 0.0
        assert err <= FX.tolerance(k, m), (k, err, FX.tolerance(k, m))
        n += 1
    assert n >= 12


# ----------------------------------------------------------------------------- round 5: the table in half2 (tinycudann's storage)
def test_tcnn_f16_table_equals_the_fp32_kernels_on_a_rounded_table():
    """module.table_f16 (MNRF_TCNN_TABLE_F16): the kernels gather from a half2 copy of the table -- 4 B per entry, what tinycudann
    stores (models/mirror_nerf_tcnn.py:39-49; SURVEY 8d's 512 B per sample) -- made from the fp32 master by mnrf_tcnn_table_half.
    A half2 -> float2 conversion is exact, so every output AND every gradient must equal, bit for bit, what the fp32-table kernels
    give on a model whose master table holds the rounded values; against the unrounded master the forward differs by f16 rounding of
    the entries (2^-11 relative).  Covers the one-launch and the level-major two-launch forward, sigma-only, the VALU kernel with the
    density-gradient normal, the encoding entry point, and the backward incl. the position gradient."""
    from mirror_nerf_amd import _lib
    from mirror_nerf_amd.mirror_nerf_tcnn import TcnnFieldFn, _offsets17
    m, _w, _cfg = _model(6.0, seed=4, table_scale=0.3)
    r, _w2, _c2 = _model(6.0, seed=4, table_scale=0.3)
    with torch.no_grad():
        r.encoder.embeddings.copy_(m.encoder.embeddings.half().float())       # the rounded twin, fp32 storage
    rays = torch.from_numpy(O.synthetic_rays(16, 16)).to(DEV)
    N, S = rays.shape[0], 40
    z = (torch.linspace(0.3, 7.5, S, device=DEV)[None] + 0.05 * torch.rand(N, S, device=DEV)).contiguous()
    m.table_f16 = True
    with torch.no_grad():
        for kw in (dict(), dict(sigma_only=True), dict(grad_normal=True), dict(want_geo=True)):
            for planes_min in (1, 1 << 62):
                m.enc_planes_min = r.enc_planes_min = planes_min
                a = m.field(N * S, rays=rays, z_vals=z, spr=S, **kw)
                b = r.field(N * S, rays=rays, z_vals=z, spr=S, **kw)
                for k in b:
                    assert torch.equal(a[k], b[k]), (kw, planes_min, k)
        del m.enc_planes_min, r.enc_planes_min
        m.table_f16 = False
        full = m.field(N * S, rays=rays, z_vals=z, spr=S)
        m.table_f16 = True
        half = m.field(N * S, rays=rays, z_vals=z, spr=S)
    d = float((full["sigma"] - half["sigma"]).abs().max())
    assert 0 < d <= 2e-3 * max(1.0, float(full["sigma"].abs().max())), d            # f16 entries: different, and close
    # the encoding entry point
    pa, pb = torch.empty(16, N * S, 2, device=DEV), torch.empty(16, N * S, 2, device=DEV)
    th, flag = m._table()
    assert flag == _lib.MNRF_TCNN_TABLE_F16 and th.dtype == torch.float16
    offs = _offsets17(m.cfg)
    _lib.check(_lib.lib().mnrf_tcnn_encode_flags(th.data_ptr(), offs, m.cfg["S"], m.cfg["H"], float(m.bound), N * S, None, 0, _lib.ptr(rays),
                                                 _lib.ptr(z), S, _lib.ptr(pa), flag, _lib.stream()), "encode f16")
    tr = r.encoder.embeddings.detach().contiguous()
    _lib.check(_lib.lib().mnrf_tcnn_encode(_lib.ptr(tr), offs, m.cfg["S"], m.cfg["H"], float(m.bound), N * S, None, 0, _lib.ptr(rays),
                                           _lib.ptr(z), S, _lib.ptr(pb), _lib.stream()), "encode")
    assert torch.equal(pa, pb)
    # backward: gradients of the master table, of the MLPs and of the positions
    x6 = torch.cat([(rays[:, None, :3] + rays[:, None, 3:6] * z[..., None]).reshape(-1, 3),
                    rays[:, None, 3:6].expand(N, S, 3).reshape(-1, 3)], 1).contiguous()
    grads = []
    for mod in (m, r):
        x = x6.clone().requires_grad_(True)
        for q in mod.parameters():
            q.grad = None
        sigma, rgb, pn, mir, normal, _geo = TcnnFieldFn.apply(mod, 1, x, None, None, None, True, mod.encoder.embeddings, *mod.mlp_params())
        (sigma.sum() + (rgb ** 2).sum() + pn[:, 0].sum() + mir.sum() + normal[:, 1].sum()).backward()
        grads.append([x.grad.clone(), mod.encoder.embeddings.grad.clone()] + [q.grad.clone() for q in mod.mlp_params()])
    assert torch.equal(grads[0][0], grads[1][0])                     # positions: no atomics, bit for bit
    for ga, gb in zip(grads[0][1:], grads[1][1:]):                   # table and MLPs: sums of atomics, equal to their rounding
        assert float((ga - gb).abs().max()) <= 2e-5 * float(gb.abs().max()) + 1e-30
    assert float(grads[0][1].abs().max()) > 0


def test_tcnn_f16_table_follows_the_master_after_an_optimizer_step():
    """The half2 copy is re-made when the fp32 master changed (a fused optimizer step bumps the generation the copy is keyed on)."""
    m, _w, _cfg = _model(6.0, seed=5, table_scale=0.3)
    m.table_f16 = True
    rays = torch.from_numpy(O.synthetic_rays(8, 8)).to(DEV)
    z = torch.linspace(0.3, 7.5, 32, device=DEV)[None].expand(rays.shape[0], 32).contiguous()
    with torch.no_grad():
        a = m.field(rays.shape[0] * 32, rays=rays, z_vals=z, spr=32)["sigma"].clone()
    opt = torch.optim.Adam([m.encoder.embeddings], lr=1e-2, fused=True)
    m.encoder.embeddings.grad = torch.ones_like(m.encoder.embeddings)
    opt.step()
    with torch.no_grad():
        b = m.field(rays.shape[0] * 32, rays=rays, z_vals=z, spr=32)["sigma"]
    assert not torch.equal(a, b)


@pytest.mark.parametrize("name", ["g17_tcnn_render_test"])
def test_g17_render_rays_with_f16_table_and_f16_mlps(name):
    """Config 5 as BASELINE words it -- half2 table, fp16 MLPs -- on the reference's own render (fixture G17) at the tolerances of
    the single-pass-f16 arithmetic: a few 1e-3 of each output's scale."""
    import mirror_nerf_amd as M
    from tests.golden import fixtures as FX
    fx = FX.Fixture(name)
    mc, _ = _g17_model(fx, "coarse__", 0)
    mf, _ = _g17_model(fx, "fine__", 1)
    for mdl in (mc, mf):
        mdl.table_f16 = mdl.mlp_f16 = True
    emb = {"xyz": M.Embedding(0), "dir": M.Embedding(0)}
    with torch.no_grad():
        got = M.render_rays({"coarse": mc, "fine": mf}, emb, torch.from_numpy(fx.inputs["rays"]).to(DEV), 64, False, 0, 0, 64,
                            32768, False, True, compute_normal=False)
    for k in ("rgb_fine", "opacity_fine", "mirror_mask_fine"):
        want = fx.outputs[k]
        assert float(np.max(np.abs(got[k].cpu().numpy() - want))) <= 5e-3 * max(1.0, float(np.abs(want).max())), k
    far = float(fx.inputs["rays"][:, 7].max())
    assert float(np.max(np.abs(got["depth_fine"].cpu().numpy() - fx.outputs["depth_fine"]))) <= 5e-3 * far
