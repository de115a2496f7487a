"""Loss reductions: oracle and HIP kernels against the golden vectors G10 captured from the
reference's losses.TotalLoss (tests/golden/make_golden_totalloss.py): loss_sum, every loss_dict entry and
d(loss_sum)/d(every input).  Tolerance: 2e-6 relative on the scalars (fp32 reductions in a different order),
1e-6 of each gradient tensor's largest entry."""
import types

import numpy as np
import pytest
import torch

from oracle import mirror_nerf_oracle as O
from tests.golden import fixtures as FX

CASES = FX.names("g10_loss_")


def _load(name):
    z = np.load(f"{FX.HERE}/{name}.npz")
    import json
    meta = json.loads(str(z["meta"]))
    inputs = {k[4:]: z[k] for k in z.files if k.startswith("in__")}
    batch = {k[7:]: z[k] for k in z.files if k.startswith("batch__")}
    plane = {k[7:]: z[k] for k in z.files if k.startswith("plane__")}
    outs = {k[5:]: z[k] for k in z.files if k.startswith("out__")}
    return meta, inputs, batch, plane, outs


def test_fixtures_present():
    assert len(CASES) == 6, CASES


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_losses(name):
    meta, inputs, batch, plane, outs = _load(name)
    hp = dict(O.LOSS_DEFAULTS)
    hp.update(meta["hp"])
    total, d = O.total_loss({k: v.copy() for k, v in inputs.items()}, batch, hp, meta["stage"], meta["epoch"], plane)
    want = {k[6:]: v for k, v in outs.items() if k.startswith("loss__")}
    assert set(d) == set(want)
    for k in want:
        assert abs(float(d[k]) - float(want[k])) <= 2e-6 * max(1.0, abs(float(want[k]))), k
    assert abs(float(total) - float(outs["loss_sum"])) <= 2e-6 * max(1.0, abs(float(outs["loss_sum"])))


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_total_loss_matches_reference(name):
    import mirror_nerf_amd as M
    meta, inputs, batch, plane, outs = _load(name)
    hp = dict(O.LOSS_DEFAULTS)
    hp.update(meta["hp"])
    dev = "cuda:0"
    tin = {k: torch.from_numpy(v.copy()).to(dev).requires_grad_(True) for k, v in inputs.items()}
    res = {k: v * 1.0 for k, v in tin.items()}        # non-leaf, as render_rays returns them
    tb = {k: torch.from_numpy(np.asarray(v).copy()).to(dev) for k, v in batch.items()}
    crit = M.get_loss(types.SimpleNamespace(**hp))
    torch.manual_seed(meta["rng_seed"])                # PlaneConsistentLoss draws from the default CPU generator
    loss_sum, loss_dict = crit(res, tb, train_geometry_stage=meta["stage"], epoch=meta["epoch"])
    want = {k[6:]: v for k, v in outs.items() if k.startswith("loss__")}
    assert set(loss_dict) == set(want), (set(loss_dict), set(want))
    for k in want:
        assert abs(float(loss_dict[k]) - float(want[k])) <= 2e-6 * max(1.0, abs(float(want[k]))), (k, float(loss_dict[k]), float(want[k]))
    assert abs(float(loss_sum) - float(outs["loss_sum"])) <= 2e-6 * max(1.0, abs(float(outs["loss_sum"])))
    loss_sum.backward()
    for k, v in tin.items():
        w = outs["grad__" + k]
        g = v.grad.cpu().numpy() if v.grad is not None else np.zeros_like(w)
        scale = max(float(np.abs(w).max()), 1e-12)
        err = float(np.abs(g - w).max())
        assert err <= 1e-6 * scale + 1e-12, f"{name}: grad {k} max-abs {err:.3e} (scale {scale:.3e})"
    if meta["stage"] and (batch["mirror_mask"] < 0).any():
        # the reference's in-place threshold of the predicted mask is reproduced
        key = "mirror_mask_fine" if "mirror_mask_fine" in res else "mirror_mask_coarse"
        m = res[key].detach().cpu().numpy()
        assert set(np.unique(m)) <= {0.0, 0.5, 1.0}


@pytest.mark.gpu
def test_hip_total_loss_full_batch_properties():
    """BASELINE config 2 training batch shape (1024 rays, 64 + 192 samples): linearity in the coefficients and
    agreement of the stored gradient with a finite difference along a random direction."""
    import mirror_nerf_amd as M
    dev = "cuda:0"
    rs = np.random.RandomState(0)
    n, sc, sf = 1024, 64, 192

    def unit(*shape):
        v = rs.normal(size=shape + (3,)).astype(np.float32)
        return v / np.linalg.norm(v, axis=-1, keepdims=True)

    res = {}
    for typ, s in (("coarse", sc), ("fine", sf)):
        res[f"rgb_{typ}"] = rs.uniform(size=(n, 3)).astype(np.float32)
        res[f"mirror_mask_{typ}"] = rs.uniform(0.01, 0.99, size=n).astype(np.float32)
        res[f"normal_dif_{typ}"] = rs.uniform(size=n).astype(np.float32)
        res[f"pred_normal_{typ}"] = unit(n, s)
        w = rs.uniform(size=(n, s)).astype(np.float32)
        res[f"weights_{typ}"] = w / w.sum(-1, keepdims=True)
        res[f"x_surface_{typ}"] = rs.normal(size=(n, 3)).astype(np.float32)
    res["normal_fine"] = unit(n, sf)
    batch = {"rgbs": rs.uniform(size=(n, 3)).astype(np.float32), "mirror_mask": (rs.uniform(size=(n, 1)) < 0.25).astype(np.float32),
             "rays": np.concatenate([rs.normal(size=(n, 3)), unit(n), np.zeros((n, 2))], 1).astype(np.float32)}
    tb = {k: torch.from_numpy(v).to(dev) for k, v in batch.items()}
    hp = dict(O.LOSS_DEFAULTS)
    want_total, want = O.total_loss({k: v.copy() for k, v in res.items()}, batch, hp, False, 5, None)

    def run(scale=1.0, shift=None):
        t = {k: torch.from_numpy(v.copy()).to(dev).requires_grad_(True) for k, v in res.items()}
        if shift is not None:
            with torch.no_grad():
                t["rgb_fine"] += shift
        h = dict(hp)
        for k in ("color_loss_weight", "normal_loss_weight", "normal_reg_loss_weight", "mirror_mask_loss_weight"):
            h[k] = hp[k] * scale
        ls, ld = M.get_loss(types.SimpleNamespace(**h))({k: v * 1.0 for k, v in t.items()}, tb, False, 5)
        return ls, ld, t

    ls, ld, t = run()
    assert abs(float(ls) - float(want_total)) <= 2e-6 * abs(float(want_total))
    for k in want:
        assert abs(float(ld[k]) - float(want[k])) <= 2e-6 * max(abs(float(want[k])), 1e-3), k
    ls2, _, _ = run(scale=2.0)
    assert abs(float(ls2) - 2 * float(ls)) <= 1e-6 * abs(float(ls2))
    ls.backward()
    d = torch.from_numpy(rs.normal(size=(n, 3)).astype(np.float32)).to(dev)
    eps = 1e-2
    lp, _, _ = run(shift=eps * d)
    lm, _, _ = run(shift=-eps * d)
    fd = (float(lp) - float(lm)) / (2 * eps)
    an = float((t["rgb_fine"].grad * d).sum())
    assert abs(fd - an) <= 2e-3 * max(abs(an), 1e-6), (fd, an)


@pytest.mark.gpu
def test_device_psnr_matches_oracle():
    """metrics.py:5-15 on the device: full 800x800 frame, with and without a per-pixel mask."""
    from mirror_nerf_amd import metrics
    rs = np.random.RandomState(3)
    a = rs.uniform(size=(640000, 3)).astype(np.float32)
    b = np.clip(a + rs.normal(scale=0.05, size=a.shape), 0, 1).astype(np.float32)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    want = float(O.psnr(a, b))
    assert abs(float(metrics.psnr(ta, tb)) - want) <= 1e-4
    assert abs(float(metrics.mse(ta, tb)) - float(np.mean((a.astype(np.float64) - b) ** 2))) <= 1e-8
    m = rs.uniform(size=640000) < 0.3
    want_m = -10 * np.log10(np.mean((a[m].astype(np.float64) - b[m]) ** 2))
    assert abs(float(metrics.psnr(ta, tb, torch.from_numpy(m).cuda())) - want_m) <= 1e-4
    assert torch.isnan(metrics.psnr(ta, tb, torch.zeros(640000, dtype=torch.bool).cuda()))


@pytest.mark.gpu
def test_training_color_mask_loss_is_the_fused_color_and_mask_terms():
    """training.color_mask_loss = the reference's ColorLoss + MirrorMaskLoss (losses.py:7-51, 175-198; both `typ`s, nn.BCELoss's
    log clamp at -100, weight 0.1) through the fused loss kernel: value and gradients against the same two terms in torch ops."""
    import torch
    from mirror_nerf_amd import training
    dev = "cuda:0"
    g = torch.Generator().manual_seed(5)
    n = 777
    res = {k: torch.rand(n, 3, generator=g).to(dev).requires_grad_(True) for k in ("rgb_coarse", "rgb_fine")}
    res.update({k: torch.rand(n, generator=g).to(dev).requires_grad_(True) for k in ("mirror_mask_coarse", "mirror_mask_fine")})
    with torch.no_grad():
        res["mirror_mask_fine"][:5] = torch.tensor([0.0, 1.0, 1e-9, 1 - 1e-9, 0.5])       # the clamp's territory
    target = torch.rand(n, 3, generator=g).to(dev)
    gt = (torch.rand(n, generator=g) < 0.3).float().to(dev)
    loss = training.color_mask_loss(res, target, gt)
    loss.backward()
    got = {k: v.grad.clone() for k, v in res.items()}
    for v in res.values():
        v.grad = None
    ref = sum(((res[k] - target) ** 2).mean() for k in ("rgb_coarse", "rgb_fine"))
    for k in ("mirror_mask_coarse", "mirror_mask_fine"):
        ref = ref + 0.1 * torch.nn.functional.binary_cross_entropy(torch.clamp(res[k], 1e-7, 1 - 1e-7), gt)      # losses.py:187-190
    ref.backward()
    assert abs(loss.item() - ref.item()) <= 1e-5 * abs(ref.item())
    for k, v in res.items():
        finite = torch.isfinite(v.grad)
        assert bool(finite.float().mean() > 0.99)
        assert float((got[k] - v.grad)[finite].abs().max()) <= 1e-5 * float(v.grad[finite].abs().max()), k
