"""GPU parity tests: the HIP path (through the C ABI) against the committed golden vectors
captured from the reference, and against the oracle on seeded inputs.

Tolerances (stated once, used everywhere below):
  * composited per-ray outputs rgb / opacity / mirror mask / predicted surface normal: 1e-4 abs
    (BASELINE.json north_star); depth, x_surface: 1e-4 relative to far = 8;
  * keys derived from the normalised autograd density gradient: floor-aware, see
    tests/golden/fixtures.py (the reference's own fp32-vs-fp64 difference on them is up to 1e-2);
  * per-sample tensors at the fine sample positions are compared only through
    order-insensitive reductions (positions are not stable at 1e-4).
In practice the HIP path agrees with the reference to ~1e-6 on the well-conditioned keys.
"""
import os

import numpy as np
import pytest
import torch

from oracle import mirror_nerf_oracle as O
from tests.golden import fixtures as FX

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
EMB_O = {"xyz": 10, "dir": 4}


@pytest.fixture(autouse=True, params=["split", "fp32"])
def precision(request):
    """Every parity test runs on both arithmetics of the field kernel: the default split-f16 tuning (fp32
    operands as hi/lo f16 pairs on the f16 matrix pipe) and the bit-exact fp32 MFMA chain.  Same tolerances."""
    from mirror_nerf_amd import mirror_nerf as MN
    old = MN.PRECISION
    MN.set_precision(request.param)
    yield request.param
    MN.set_precision(old)


def _M():
    import mirror_nerf_amd as M
    return M


def _module(sd):
    M = _M()
    m = M.MirrorNeRF(in_channels_xyz=63, in_channels_dir=27, predict_normal="normal_net.0.weight" in sd,
                     predict_mirror_mask="is_mirror_net.0.weight" in sd)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.to(DEV)


def _emb():
    M = _M()
    return {"xyz": M.Embedding(10), "dir": M.Embedding(4)}


def _np(d):
    return {k: v.detach().cpu().numpy() for k, v in d.items()}


def _cmp(name, got, want_dict, meta, skip=(), base=1e-4):
    for k, want in want_dict.items():
        if k in skip:
            continue
        assert k in got, f"{name}: missing key {k}"
        g = got[k]
        assert g.shape == want.shape, (name, k, g.shape, want.shape)
        tol = FX.tolerance(k, meta, base)
        err = float(np.max(np.abs(g.astype(np.float64) - want))) if want.size else 0.0
        assert err <= tol, f"{name}:{k} max-abs {err:.3e} > {tol:.1e}"
    extra = set(got) - set(want_dict) - set(skip)
    assert not extra or all(k.startswith(("pred_normal_", "normal_")) for k in extra), extra


# --------------------------------------------------------------------------- a1
def test_embedding_golden():
    fx = FX.Fixture("g1_embedding")
    M = _M()
    x = torch.from_numpy(fx.inputs["x"]).to(DEV)
    d = torch.from_numpy(fx.inputs["d"]).to(DEV)
    assert np.max(np.abs(M.Embedding(10)(x).cpu().numpy() - fx.outputs["e10"])) <= 2e-6
    assert np.max(np.abs(M.Embedding(4)(d).cpu().numpy() - fx.outputs["e4"])) <= 2e-6
    assert np.array_equal(M.Embedding(0)(x).cpu().numpy(), fx.outputs["e0"])
    assert M.Embedding(4)(x[:0]).shape == (0, 27)


# --------------------------------------------------------------------------- a2-a5
def test_field_golden():
    fx = FX.Fixture("g2_field")
    m = _module(fx.state_dicts()[0])
    e = _emb()
    x30 = torch.from_numpy(fx.inputs["x30"]).to(DEV)
    full = _np(m(x30, compute_normal=True, sigma_only=False, embedding_xyz=e["xyz"], embedding_dir=e["dir"]))
    sonly = _np(m(x30[:, :3].contiguous(), compute_normal=False, sigma_only=True, embedding_xyz=e["xyz"],
                  embedding_dir=e["dir"]))
    assert full["sigma"].shape == (512, 1) and full["is_mirror"].shape == (512, 1)
    for k, want in fx.outputs.items():
        mode, key = k.split("__")
        got = (full if mode == "full" else sonly)[key]
        if key == "geo_feat":
            got = got[:, :8]
        err = float(np.max(np.abs(got - want)))
        assert err <= 2e-5, f"{k}: {err:.3e}"


@pytest.mark.parametrize("B", [1, 15, 16, 17, 127, 128, 129, 1000])
def test_field_ragged_sizes_vs_oracle(B):
    """Tile edges: the kernel works on 128-sample tiles of 4 waves x 2 x 16."""
    from tests.golden import weights as GW
    sd = GW.apply_tweaks(GW.make_state_dict(3, 1)[0], GW.OPAQUE)
    m = _module(sd)
    e = _emb()
    rs = np.random.RandomState(B)
    xyz = rs.uniform(-4, 4, (B, 3)).astype(np.float32)
    d = rs.normal(size=(B, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    x30 = np.concatenate([xyz, O.embedding(d, 4)], 1)
    want = O.field_forward(sd, x30, False, True)
    got = _np(m(torch.from_numpy(x30).to(DEV), compute_normal=True, embedding_xyz=e["xyz"], embedding_dir=e["dir"]))
    for k in ("sigma", "rgb", "pred_normal", "is_mirror", "geo_feat"):
        err = float(np.max(np.abs(got[k] - want[k])))
        assert err <= 3e-5, (k, err)
    # normal: compare where the gradient is well conditioned
    dn = np.abs(got["normal"] - want["normal"]).max(-1)
    assert np.median(dn) <= 1e-4


def test_field_empty():
    from tests.golden import weights as GW
    m = _module(GW.make_state_dict(3, 1)[0])
    e = _emb()
    out = m(torch.zeros(0, 30, device=DEV), compute_normal=False, embedding_xyz=e["xyz"], embedding_dir=e["dir"])
    assert out["sigma"].shape == (0, 1) and out["rgb"].shape == (0, 3)


# --------------------------------------------------------------------------- a6-a11
def _render_fixture(name):
    fx = FX.Fixture(name)
    m = fx.meta
    sds = fx.state_dicts()
    models = {"coarse": _module(sds[0])}
    if m["N_importance"] > 0 and not m["kwargs"].get("only_one_field", False):
        models["fine"] = _module(sds[1])
    kw = dict(m["kwargs"])
    for k in ("perturb_rand", "noise_coarse", "noise_fine", "u"):
        if k in fx.inputs:
            kw["_" + k] = torch.from_numpy(fx.inputs[k]).to(DEV)
    rays = torch.from_numpy(fx.inputs["rays"]).to(DEV)
    got = _np(_M().render_rays(models, _emb(), rays, m["N_samples"], m["use_disp"], m["perturb"], m["noise_std"],
                               m["N_importance"], m["chunk"], m["white_back"], m["test_time"], **kw))
    return fx, got


@pytest.mark.parametrize("name", FX.names("g3_") + FX.names("g4_") + FX.names("g5_"))
def test_render_rays_golden(name):
    fx, got = _render_fixture(name)
    skip = list(FX.PER_SAMPLE_FINE)
    if fx.meta["kwargs"].get("only_one_field") and fx.meta["kwargs"].get("current_epoch", 0) > 2:
        skip += ["z_vals_coarse", "weights_coarse", "pred_normal_coarse", "normal_coarse"]
    skip += ["normal_coarse", "normal_fine"]          # per-sample normalised gradients (noise-dominated)
    _cmp(name, got, fx.outputs, fx.meta, skip=skip)
    for typ in ("coarse", "fine"):
        if f"weights_{typ}" in got and f"opacity_{typ}" in got:
            assert np.max(np.abs(got[f"weights_{typ}"].sum(1) - got[f"opacity_{typ}"])) <= 1e-5
            assert np.all(np.diff(got[f"z_vals_{typ}"], axis=1) >= 0)
    if "weights_fine" in fx.outputs and name != "g4_fine_rawinit":
        assert np.max(np.abs(got["weights_fine"].sum(1) - fx.outputs["weights_fine"].sum(1))) <= 1e-4
        # the multiset of fine depths moves by at most a few 1e-4 (bin flips), never wholesale
        assert np.max(np.abs(got["z_vals_fine"] - fx.outputs["z_vals_fine"])) <= 5e-3


def _normal_error_ladder(k, e_hip, e_ref):
    """Per-sample gradient normals against the fp64 truth, compared as error DISTRIBUTIONS (mean and
    median alone do not see the tail): every quantile of the HIP error up to 99.9 % within twice the reference's fp32 run's,
    and no more samples beyond 1e-3 / 1e-2 / 1e-1 than twice the reference's count (+ 2 samples).  The handful of samples
    with a vanishing density gradient that carry O(1) errors in ANY fp32 run sit above the ladder in both."""
    eh, er = np.linalg.norm(e_hip.reshape(-1, 3), axis=1), np.linalg.norm(e_ref.reshape(-1, 3), axis=1)
    assert np.abs(e_hip).mean() <= 2.0 * np.abs(e_ref).mean() + 1e-6, (k, np.abs(e_hip).mean(), np.abs(e_ref).mean())
    assert np.median(np.abs(e_hip)) <= 2.0 * np.median(np.abs(e_ref)) + 1e-6, k
    ladder = {}
    for q in (50.0, 90.0, 99.0, 99.9):
        a, b = float(np.percentile(eh, q)), float(np.percentile(er, q))
        ladder[q] = (a, b)
        assert a <= 2.0 * b + 1e-6, (k, "quantile", q, a, b)
    for lim in (1e-3, 1e-2, 1e-1):
        a, b = int((eh > lim).sum()), int((er > lim).sum())
        ladder[lim] = (a, b)
        assert a <= 2 * b + 2, (k, "samples beyond", lim, a, b)
    return ladder



@pytest.mark.parametrize("base", ["g4_fine_train", "g11_trained_render_train"])
def test_fine_pass_as_close_to_the_fp64_truth_as_the_reference(base, precision):
    """Fixtures G14 of the FINE pass (tests/golden/make_golden_truth64.py truth_fine): the reference in float64 at the fine
    depths its own fp32 run drew -- on random-init weights (g4_fine_train) and on the trained pair (g11_trained_render_train).
    With the same depths injected (`_z_fine`; inverse-CDF bin flips are a property of the resampling, pinned by the G4/G11
    fixture tests) the HIP result has to be as close to the TRUTH as the reference's fp32 run is: max-abs error within twice
    the reference's plus roundoff of the key's scale (2e-6 for the fp32 chain; 1e-5 for the split arithmetic, whose field
    outputs sit 3e-6 from the fp32 chain's before compositing), mean / median for the per-sample gradient normals.  This is
    the justification of the 2e-2 / 4 x floor allowances on the fine pass and on trained weights."""
    fx = FX.Fixture(base)
    m = fx.meta
    sds = fx.state_dicts()
    models = {"coarse": _module(sds[0]), "fine": _module(sds[1])}
    rays = torch.from_numpy(fx.inputs["rays"]).to(DEV)
    got = _np(_M().render_rays(models, _emb(), rays, m["N_samples"], m["use_disp"], m["perturb"], m["noise_std"], m["N_importance"],
                               m["chunk"], m["white_back"], m["test_time"],
                               _z_fine=torch.from_numpy(fx.outputs["z_vals_fine"]).to(DEV), **m["kwargs"]))
    z = np.load(os.path.join(os.path.dirname(FX.__file__), f"g14_truth64_{base}.npz"))
    slack = 2e-6 if precision == "fp32" else 1e-5
    checked, report = 0, {}
    for f in z.files:
        if not f.startswith("out64__"):
            continue
        k = f[7:]
        truth = z[f]
        ref32 = fx.outputs[k].astype(np.float64)
        hip = got[k].astype(np.float64)
        scale = 8.0 if k.startswith(("depth", "x_surface", "z_vals")) else 1.0
        e_ref, e_hip = np.abs(ref32 - truth), np.abs(hip - truth)
        report[k] = (float(e_hip.max()), float(e_ref.max()))
        if k == "normal_fine":
            print("normal_fine error ladder (hip, reference fp32):", _normal_error_ladder(k, hip - truth, ref32 - truth))
        else:
            assert e_hip.max() <= 2.0 * e_ref.max() + slack * scale, (k, e_hip.max(), e_ref.max())
        checked += 1
    assert checked >= 9
    print(f"G14 fine {base} [{precision}] max |err| vs fp64 truth (hip, reference fp32):",
          {k: (f"{a:.1e}", f"{b:.1e}") for k, (a, b) in report.items()})


@pytest.mark.parametrize("base", ["g6_train_gt_compact", "g6_train_pred_straddle"])
def test_recursion_primary_level_as_close_to_the_fp64_truth_as_the_reference(base, precision):
    """Fixtures G14 for the RECURSION fixtures (tests/golden/make_golden_truth64.py truth_recursion_level0; VERDICT r4 weak #2: the
    G6 comparisons of surface_normal_grad_* / normal_dif_* ride on the 2e-2 allowance of the noise-dominated keys).  Those keys are
    the primary render's (train.py:132-145; the recursion only re-blends rgb_*), so their truth is the reference's render_rays in
    float64 on the fixture's rays and weights at the fine depths its fp32 run drew.  The HIP render of the same call has to be as
    close to that truth as the reference's fp32 values in the fixture are: max-abs error within twice the reference's plus
    roundoff (2e-6 fp32 chain, 1e-5 split arithmetic).  The reference's own fp32 error on these keys is 1e-3 .. 7e-3."""
    fx = FX.Fixture(base)
    hp = fx.meta["hp"]
    sds = fx.state_dicts()
    models = {"coarse": _module(sds[0]), "fine": _module(sds[1])}
    rays = torch.from_numpy(fx.inputs["rays"]).to(DEV)
    got = _np(_M().render_rays(models, _emb(), rays, hp["N_samples"], hp["use_disp"], hp["perturb"], hp["noise_std"], hp["N_importance"],
                               hp["chunk"], False, compute_normal=hp["trace_secondary_rays"],
                               _z_fine=torch.from_numpy(fx.outputs["z_vals_fine"]).to(DEV)))
    z = np.load(os.path.join(os.path.dirname(FX.__file__), f"g14_truth64_{base}.npz"))
    slack = 2e-6 if precision == "fp32" else 1e-5
    thresholded = fx.meta.get("gt_mode") == "invalid"      # train.py:155-166: the fixture's predicted masks are the hard ones
    checked, report = 0, {}
    for f in z.files:
        if not f.startswith("out64__"):
            continue
        k = f[7:]
        if thresholded and k.startswith("mirror_mask"):
            continue
        truth = z[f]
        ref32 = fx.outputs[k + "_direct" if k in ("rgb_coarse", "rgb_fine") else k].astype(np.float64)
        hip = got[k].astype(np.float64)
        scale = 8.0 if k.startswith(("depth", "x_surface", "z_vals")) else 1.0
        e_ref, e_hip = np.abs(ref32 - truth), np.abs(hip - truth)
        report[k] = (float(e_hip.max()), float(e_ref.max()))
        assert e_hip.max() <= 2.0 * e_ref.max() + slack * scale, (k, e_hip.max(), e_ref.max())
        checked += 1
    assert checked >= 18
    print(f"G14 recursion primary level {base} [{precision}] max |err| vs fp64 truth (hip, reference fp32):",
          {k: (f"{a:.1e}", f"{b:.1e}") for k, (a, b) in report.items()})


# --------------------------------------------------------------------------- G14: against the reference in float64
def test_as_close_to_the_fp64_truth_as_the_reference(precision):
    """Fixture G14 (tests/golden/make_golden_truth64.py): the reference evaluated in float64 on the inputs of
    g3_coarse64_train.  The keys derived from the normalised autograd density gradient are noise-dominated in the
    reference's own fp32 run (max-abs error against fp64: 8e-3 on surface_normal_grad, 1e-2 on normal_dif, 0.6 on single
    samples of normal_coarse), which is why the plain fixture comparison carries a 2e-2 floor for them.  Here the sharper
    question: the HIP result has to be as close to the TRUTH as the reference's fp32 result is -- max-abs error within twice
    the reference's (plus fp32 roundoff of the key's scale), mean-abs error of the per-sample normals within twice the
    reference's -- for both arithmetics."""
    fx, got = _render_fixture("g3_coarse64_train")
    z = np.load(os.path.join(os.path.dirname(FX.__file__), "g14_truth64_g3_coarse64_train.npz"))
    checked = 0
    for f in z.files:
        if not f.startswith("out64__"):
            continue
        k = f[7:]
        truth = z[f]
        ref32 = fx.outputs[k].astype(np.float64)
        hip = got[k].astype(np.float64)
        scale = 8.0 if k.startswith(("depth", "x_surface", "z_vals")) else 1.0
        e_ref, e_hip = np.abs(ref32 - truth), np.abs(hip - truth)
        if k == "normal_coarse":       # per sample: a few samples with a vanishing gradient carry errors of O(1) in any fp32 run
            print("normal_coarse error ladder (hip, reference fp32):", _normal_error_ladder(k, hip - truth, ref32 - truth))
        else:
            assert e_hip.max() <= 2.0 * e_ref.max() + 2e-6 * scale, (k, e_hip.max(), e_ref.max())
        checked += 1
    assert checked >= 10


# --------------------------------------------------------------------------- G13: without the optional heads
@pytest.mark.parametrize("name", [n for n in FX.names("g13_") if not n.endswith("_eval")])
def test_render_rays_without_optional_heads_golden(name):
    """models/mirror_nerf.py:80-99: predict_normal=False and/or predict_mirror_mask=False (plain NeRF for
    extract_color_mesh.py, ablations): same kernels, absent heads packed as zeros, their keys absent from the dict."""
    fx = FX.Fixture(name)
    m = fx.meta
    from tests.golden import weights as GW
    sds = GW.make_state_dict(m["seed"], 2, predict_normal=m["predict_normal"], predict_mirror_mask=m["predict_mirror_mask"])
    for sd, c in zip(sds, m["checksum"]):
        GW.apply_tweaks(sd, m["tweaks"])
        assert abs(GW.checksum(sd) - c) <= 1e-9 * max(1.0, abs(c))
    models = {"coarse": _module(sds[0]), "fine": _module(sds[1])}
    rays = torch.from_numpy(fx.inputs["rays"]).to(DEV)
    got = _np(_M().render_rays(models, _emb(), rays, 64, False, 0, 0, 64, 32768, False, m["test_time"], **m["kwargs"]))
    assert set(got) - set(PER_SAMPLE_OK) == set(fx.outputs) - set(PER_SAMPLE_OK), set(got) ^ set(fx.outputs)
    _cmp(name, got, fx.outputs, m, skip=list(FX.PER_SAMPLE_FINE) + ["normal_coarse", "normal_fine"])


def test_eval_without_normal_head_uses_the_density_gradient_normal():
    """predict_normal=False: eval.py:147-148 turns compute_normal on and reflects about the composited density-gradient
    normal (eval.py:338-360) -- a noise-dominated quantity in fp32 (floors measured per key in the fixture)."""
    fx = FX.Fixture("g13_no_normal_head_eval")
    m = fx.meta
    from tests.golden import weights as GW
    sds = GW.make_state_dict(m["seed"], 2, predict_normal=False, predict_mirror_mask=True)
    for sd in sds:
        GW.apply_tweaks(sd, m["tweaks"])
    models = {"coarse": _module(sds[0]), "fine": _module(sds[1])}
    got = _np(_M().batched_inference(models, _emb(), torch.from_numpy(fx.inputs["rays"]).to(DEV), 64, 64, False, 32768,
                                     args=m["args"], trace_secondary_rays=True))
    assert "surface_normal_fine" not in got and "surface_normal_grad_fine" in got
    meta = dict(m, floor={k: max(v, 5e-3 if k in ("reflect_direction",) else 0.0) for k, v in m["floor"].items()})
    _cmp("g13_no_normal_head_eval", got, fx.outputs, meta, skip=FX.PER_SAMPLE_FINE)


PER_SAMPLE_OK = ("pred_normal_coarse", "pred_normal_fine", "normal_coarse", "normal_fine")


# --------------------------------------------------------------------------- G11: trained weights
def _cmp_trained(name, got, fx, skip=()):
    """Parity on TRAINED weights (fixtures G11).  There the reference is not stable at 1e-4 against ITSELF: its fp32 and
    fp64 runs put a few fine samples into different inverse-CDF bins, and on those rays the composited outputs move by up
    to 1e-2 while every other ray agrees to 1e-6 (both numbers are measured per fixture: meta.floor, meta.floor_frac).
    Two conditions, both relative to that measured noise: (1) max-abs <= max(1e-4 [8e-4 depth-like], 4 x floor);
    (2) the FRACTION of rays that differ by more than the plain 1e-4 / 8e-4 bar is at most twice the reference's own
    fraction plus two rays -- so the bulk of the rays still has to meet the north-star bar exactly."""
    meta = fx.meta
    n_checked = 0
    for k, want in fx.outputs.items():
        if k in skip or k not in got:
            continue
        g = got[k].astype(np.float64)
        assert g.shape == want.shape, (name, k)
        tol = FX.tolerance(k, meta)
        d = np.abs(g - want).reshape(want.shape[0], -1).max(1) if want.size else np.zeros(0)
        assert d.max(initial=0.0) <= tol, f"{name}:{k} max-abs {d.max():.3e} > {tol:.1e}"
        bar = 8e-4 if k.startswith(("depth", "x_surface", "z_vals")) else 1e-4
        if k in FX.GRAD_NORMAL_KEYS:
            continue
        frac = float((d > bar).mean()) if d.size else 0.0
        allowed = 2.0 * meta.get("floor_frac", {}).get(k, 0.0) + 2.0 / max(1, d.size)
        assert frac <= allowed, f"{name}:{k} {frac:.4f} of the rays off by more than {bar:.0e} (reference fp32 vs fp64: {allowed:.4f} allowed)"
        n_checked += 1
    assert n_checked >= 4, (name, n_checked)


@pytest.mark.parametrize("name", [n for n in FX.names("g11_") if "_render_" in n])
def test_trained_weights_render_golden(name):
    fx, got = _render_fixture(name)
    skip = list(FX.PER_SAMPLE_FINE) + ["normal_coarse", "normal_fine", "pred_normal_coarse"]
    _cmp_trained(name, got, fx, skip)
    for typ in ("coarse", "fine"):
        assert np.max(np.abs(got[f"weights_{typ}"].sum(1) - got[f"opacity_{typ}"])) <= 1e-5
        assert np.all(np.diff(got[f"z_vals_{typ}"], axis=1) >= 0)


@pytest.mark.parametrize("name", [n for n in FX.names("g11_") if "_eval_" in n])
def test_trained_weights_eval_golden(name):
    fx = FX.Fixture(name)
    m = fx.meta
    sds = fx.state_dicts()
    models = {"coarse": _module(sds[0]), "fine": _module(sds[1])}
    got = _np(_M().batched_inference(models, _emb(), torch.from_numpy(fx.inputs["rays"]).to(DEV), m["N_samples"],
                                     m["N_importance"], False, m["chunk"], args=m["args"], trace_secondary_rays=True,
                                     normal_noise_std=m["args"]["normal_noise_std"]))
    _cmp_trained(name, got, fx, FX.PER_SAMPLE_FINE)


@pytest.mark.parametrize("name", [n for n in FX.names("g11_") if n.endswith("_psnr")])
def test_trained_weights_psnr_within_a_tenth_of_a_db(name):
    """BASELINE north star: PSNR within 0.1 dB of the reference on the synthetic mirror scene.  The held-out view of the
    analytic scene rendered by the reference (captured) and by the HIP path, both scored against the analytic ground
    truth with the device metric."""
    from mirror_nerf_amd import metrics
    fx = FX.Fixture(name)
    m = fx.meta
    sds = fx.state_dicts()
    models = {"coarse": _module(sds[0]), "fine": _module(sds[1])}
    rays = torch.from_numpy(fx.inputs["rays"]).to(DEV)
    out = _M().batched_inference(models, _emb(), rays, m["N_samples"], m["N_importance"], False, 32768, args=m["args"],
                                 trace_secondary_rays=True, to_cpu=False)
    gt = torch.from_numpy(fx.inputs["gt_rgb"]).to(DEV)
    psnr = float(metrics.psnr(out["rgb_fine"], gt))
    assert abs(psnr - m["psnr_ref"]) <= 0.1, (psnr, m["psnr_ref"])
    got = _np({k: out[k] for k in fx.outputs})
    # and the HIP image is, pixel for pixel, as close to the reference's as the reference's fp64 run is
    _cmp_trained(name, got, fx)
    acc = float(((got["mirror_mask_fine"] > 0.5) == (fx.inputs["gt_mask"] > 0.5)).mean())
    assert abs(acc - m["mask_accuracy_ref"]) <= 2.0 / got["mirror_mask_fine"].size + 1e-9


def test_render_rays_empty_and_single():
    from tests.golden import weights as GW
    sds = GW.make_state_dict(0, 2)
    models = {"coarse": _module(sds[0]), "fine": _module(sds[1])}
    M = _M()
    r0 = M.render_rays(models, _emb(), torch.zeros(0, 8, device=DEV), 64, False, 0, 0, 128, test_time=True,
                       compute_normal=False)
    assert r0["rgb_fine"].shape == (0, 3) and r0["weights_coarse"].shape == (0, 64)
    rays = torch.from_numpy(O.synthetic_rays(4, 4)[5:6]).to(DEV)
    r1 = _np(M.render_rays(models, _emb(), rays, 64, False, 0, 0, 128, test_time=True, compute_normal=False))
    want = O.render_rays({"coarse": sds[0], "fine": sds[1]}, EMB_O, rays.cpu().numpy(), 64, False, 0, 0, 128,
                         test_time=True, compute_normal=False)
    assert np.max(np.abs(r1["rgb_fine"] - want["rgb_fine"])) <= 1e-4


# --------------------------------------------------------------------------- kernels one by one
def test_sample_fine_vs_oracle_given_same_weights():
    """With identical coarse weights the inverse-CDF + sort kernel reproduces the oracle's depths.
    cdf is accumulated in double like ATen's cumsum; what remains is the 1-ulp freedom of the
    weight sum, which can (a) move a sample inside a near-empty bin (denominator ~1e-5) by a few
    1e-5 and (b) flip the u = 1.0 sample between the last two bins when cdf[-1] straddles 1.0
    (SURVEY 8a hazard 6) -- both are properties of the reference algorithm itself."""
    rs = np.random.RandomState(11)
    N, S, NI = 300, 64, 128
    z = np.sort(rs.uniform(0.05, 8, (N, S)).astype(np.float32), 1)
    # well-conditioned bins: pdf >= 0.01, so a 1-ulp change of the cdf moves a sample by < 2e-6
    w = rs.uniform(0.5, 1, (N, S)).astype(np.float32)
    w[5] = 0          # empty ray: cdf from eps only
    mid = 0.5 * (z[:, :-1] + z[:, 1:])
    t = lambda a: torch.from_numpy(a).to(DEV)  # noqa: E731
    # random u strictly inside (0, 1): no knife edge, expect agreement to rounding
    u = rs.uniform(0.001, 0.999, (N, NI)).astype(np.float32)
    want = np.sort(np.concatenate([z, O.sample_pdf(mid, w[:, 1:-1], NI, det=False, u=u)], -1), -1)
    got = _M().sample_pdf(t(z), t(w), NI, det=False, u=t(u)).cpu().numpy()
    assert got.shape == (N, S + NI)
    assert np.max(np.abs(got - want)) <= 1e-5
    # deterministic u = linspace(0,1): every sample but (possibly) the u = 1.0 one agrees
    want = np.sort(np.concatenate([z, O.sample_pdf(mid, w[:, 1:-1], NI, det=True)], -1), -1)
    got = _M().sample_pdf(t(z), t(w), NI, det=True).cpu().numpy()
    bad = np.abs(got - want) > 2e-5
    assert bad.sum(1).max() <= 2 and np.all(np.diff(got, axis=1) >= 0)
    # stress: eighth-power weights (bins whose pdf is ~eps), one dominant bin
    w2 = (rs.uniform(0, 1, (N, S)) ** 8).astype(np.float32)
    w2[6, 10] = 1.0
    want = np.sort(np.concatenate([z, O.sample_pdf(mid, w2[:, 1:-1], NI, det=False, u=u)], -1), -1)
    got = _M().sample_pdf(t(z), t(w2), NI, det=False, u=t(u)).cpu().numpy()
    assert np.max(np.abs(got - want)) <= 2e-3 and np.median(np.abs(got - want)) <= 1e-6
    # the coarse depths are always part of the output
    for r in (0, 5, 6, 299):
        assert np.isin(z[r], got[r]).all()


def test_reflect_compact_and_blend():
    from mirror_nerf_amd import recursion as R
    rs = np.random.RandomState(5)
    N = 2500   # > 2 passes of the 1024-thread compaction loop, ragged tail
    rays = rs.normal(size=(N, 8)).astype(np.float32)
    xs = rs.normal(size=(N, 3)).astype(np.float32)
    nrm = rs.normal(size=(N, 3)).astype(np.float32)
    nrm[7] = 0  # degenerate normal: eps clamp
    mask = (rs.uniform(size=N) < 0.3).astype(np.float32)
    mask[11] = 0.5
    t = lambda a: torch.from_numpy(a).to(DEV)  # noqa: E731
    sec, index, rdir = R._reflect(t(rays), t(xs), t(nrm), t(mask), True)
    want_dir, _, _ = O.reflect(rays[:, 3:6], nrm)
    sel = mask != 0
    assert index.cpu().numpy().tolist() == np.nonzero(sel)[0].tolist()
    assert np.max(np.abs(rdir.cpu().numpy() - want_dir)) <= 1e-5
    s = sec.cpu().numpy()
    assert np.array_equal(s[:, 0:3], xs[sel]) and np.all(s[:, 6] == np.float32(0.1))
    assert np.array_equal(s[:, 7], rays[sel, 7])
    assert np.max(np.abs(s[:, 3:6] - want_dir[sel])) <= 1e-5
    sec_all, idx_none, _ = R._reflect(t(rays), t(xs), t(nrm), t(mask), False)
    assert idx_none is None and sec_all.shape == (N, 8)
    # blend / scatter
    base = rs.uniform(size=(N, 3)).astype(np.float32)
    secrgb = rs.uniform(size=(int(sel.sum()), 3)).astype(np.float32)
    out, refl = R._blend(t(base), t(secrgb), index, t(mask), True)
    part = base.copy()
    part[sel] = secrgb
    want = mask[:, None] * part + (1 - mask[:, None]) * base
    assert np.max(np.abs(out.cpu().numpy() - want)) <= 1e-6
    rr = np.zeros_like(base)
    rr[sel] = secrgb
    assert np.array_equal(refl.cpu().numpy(), rr)
    # nothing selected
    sec0, idx0, _ = R._reflect(t(rays), t(xs), t(nrm), t(np.zeros(N, np.float32)), True)
    assert sec0.shape[0] == 0 and idx0.shape[0] == 0


def test_threshold_mask_leaves_half_untouched():
    from mirror_nerf_amd import recursion as R
    m = torch.tensor([0.2, 0.5, 0.7, 0.4999, 0.5001], device=DEV)
    assert R._threshold_(m) is True
    assert m.cpu().tolist() == [0.0, 0.5, 1.0, 0.0, 1.0]
    z = torch.tensor([0.1, 0.3], device=DEV)
    assert R._threshold_(z) is False


# --------------------------------------------------------------------------- a12
@pytest.mark.parametrize("name", FX.names("g6_"))
def test_recursion_train_golden(name):
    from types import SimpleNamespace
    fx = FX.Fixture(name)
    sds = fx.state_dicts()
    hp = dict(fx.meta["hp"])
    hp.update(N_emb_xyz=10, N_emb_dir=4, predict_normal=True, predict_mirror_mask=True, model_type="nerf")
    system = _M().NeRFSystem(SimpleNamespace(**hp))
    system.nerf_coarse.load_state_dict({k: torch.from_numpy(v) for k, v in sds[0].items()})
    system.nerf_fine.load_state_dict({k: torch.from_numpy(v) for k, v in sds[1].items()})
    system.to(DEV)
    rays = torch.from_numpy(fx.inputs["rays"]).to(DEV)
    extra = {"mirror_mask": torch.from_numpy(fx.inputs["gt_mask"].copy()).to(DEV), "is_eval": fx.meta["is_eval"],
             "train_geometry_stage": False}
    got = _np(system(rays, extra))
    _cmp(name, got, fx.outputs, fx.meta, skip=FX.PER_SAMPLE_FINE)


# --------------------------------------------------------------------------- a13 / a14
@pytest.mark.parametrize("name", FX.names("g7_") + FX.names("g8_") + FX.names("g8b_"))
def test_recursion_eval_golden(name):
    fx = FX.Fixture(name)
    m = fx.meta
    sds = fx.state_dicts()
    models = {"coarse": _module(sds[0]), "fine": _module(sds[1])}
    n_noise = len([k for k in fx.inputs if k.startswith("normal_noise_")])
    noise = iter([torch.from_numpy(fx.inputs[f"normal_noise_{i}"]) for i in range(n_noise)])
    got = _np(_M().batched_inference(models, _emb(), torch.from_numpy(fx.inputs["rays"]).to(DEV), m["N_samples"],
                                     m["N_importance"], False, m["chunk"], args=m["args"],
                                     trace_secondary_rays=True, normal_noise_std=m["args"]["normal_noise_std"],
                                     _normal_noise=noise))
    _cmp(name, got, fx.outputs, m, skip=FX.PER_SAMPLE_FINE)


@pytest.mark.parametrize("name", FX.names("g15_c1_"))
def test_config1_through_the_recursion_golden(name):
    """G15: BASELINE config 1 -- N_importance = 0, one bounce -- through NeRFSystem.forward and batched_inference
    (`select_type = "coarse"`: train.py:147-151, eval.py:132-172; only render_rays was pinned for it before)."""
    from types import SimpleNamespace
    fx = FX.Fixture(name)
    m = fx.meta
    sd = fx.state_dicts()[0]
    rays = torch.from_numpy(fx.inputs["rays"]).to(DEV)
    if "_train_" in name:
        hp = dict(m["hp"])
        hp.update(N_emb_xyz=10, N_emb_dir=4, predict_normal=True, predict_mirror_mask=True, model_type="nerf")
        system = _M().NeRFSystem(SimpleNamespace(**hp))
        assert not hasattr(system, "nerf_fine")
        system.nerf_coarse.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        system.to(DEV)
        extra = {"mirror_mask": torch.from_numpy(fx.inputs["gt_mask"].copy()).to(DEV), "is_eval": m["is_eval"],
                 "train_geometry_stage": False}
        got = _np(system(rays, extra))
    else:
        got = _np(_M().batched_inference({"coarse": _module(sd)}, _emb(), rays, m["N_samples"], 0, False, m["chunk"],
                                         args=m["args"], trace_secondary_rays=True, normal_noise_std=0))
    assert not any(k.endswith("_fine") for k in got)
    _cmp(name, got, fx.outputs, m)


def test_roughness_jitters_batched_equal_one_by_one():
    """Config 4 (run.sh:187-188): the trace_ray_times jittered reflections of a level rendered in groups through one
    recursion call (batch_jitter, the production path) give the same colours as the reference's one-by-one loop, given
    the same draws (one bounce: no nested draws to interleave)."""
    fx = FX.Fixture("g8_rough_allmirror")
    m = fx.meta
    sds = fx.state_dicts()
    models = {"coarse": _module(sds[0]), "fine": _module(sds[1])}
    args = dict(m["args"], trace_ray_times=7)
    rays = torch.from_numpy(fx.inputs["rays"]).to(DEV)
    rs = np.random.RandomState(3)
    draws = [torch.from_numpy(rs.normal(size=(rays.shape[0], 3)).astype(np.float32)) for _ in range(8)]
    M = _M()
    from mirror_nerf_amd import recursion as R
    outs = []
    for batch, group in ((False, None), (True, 262144), (True, 3 * rays.shape[0])):    # one by one; one group; groups of 3
        old = R.JITTER_RAYS
        if group:
            R.JITTER_RAYS = group
        try:
            outs.append(M.batched_inference(models, _emb(), rays, m["N_samples"], m["N_importance"], False, m["chunk"], args=args,
                                            trace_secondary_rays=True, normal_noise_std=args["normal_noise_std"],
                                            _normal_noise=iter(draws), batch_jitter=batch, to_cpu=False))
        finally:
            R.JITTER_RAYS = old
    for o in outs[1:]:
        assert torch.equal(o["rgb_fine"], outs[0]["rgb_fine"])
        assert torch.equal(o["rgb_fine_reflect"], outs[0]["rgb_fine_reflect"])


# --------------------------------------------------------------------------- full-size properties
def test_full_size_chunk_properties():
    """One 32768-ray chunk of the 800x800 configuration (64 coarse + 128 fine): size-independent
    invariants, plus agreement with the oracle on a strided subset of the same rays."""
    from tests.golden import weights as GW
    sds = [GW.apply_tweaks(sd, GW.OPAQUE) for sd in GW.make_state_dict(0, 2)]
    models = {"coarse": _module(sds[0]), "fine": _module(sds[1])}
    rays_np = O.synthetic_rays(800, 800)[300 * 800: 300 * 800 + 32768]
    rays = torch.from_numpy(rays_np).to(DEV)
    r = _M().render_rays(models, _emb(), rays, 64, False, 0, 0, 128, test_time=True, compute_normal=False)
    w, op, z = r["weights_fine"], r["opacity_fine"], r["z_vals_fine"]
    assert w.shape == (32768, 192)
    assert torch.isfinite(r["rgb_fine"]).all() and torch.isfinite(r["depth_fine"]).all()
    assert float((w.sum(1) - op).abs().max()) <= 1e-5
    assert float(op.max()) <= 1 + 1e-5 and float(w.min()) >= 0
    assert bool((z[:, 1:] >= z[:, :-1]).all())
    assert float(r["rgb_fine"].min()) >= 0 and float(r["rgb_fine"].max()) <= 1 + 1e-5
    # rendering is per-ray: a strided subset rendered alone gives the same numbers (chunk independence)
    sub = rays[::997].contiguous()
    r2 = _M().render_rays(models, _emb(), sub, 64, False, 0, 0, 128, test_time=True, compute_normal=False)
    assert float((r2["rgb_fine"] - r["rgb_fine"][::997]).abs().max()) <= 1e-6
    want = O.render_rays({"coarse": sds[0], "fine": sds[1]}, EMB_O, rays_np[::997], 64, False, 0, 0, 128,
                         test_time=True, compute_normal=False)
    got = _np(r2)
    for k in ("rgb_fine", "opacity_fine", "mirror_mask_fine", "surface_normal_fine"):
        assert np.max(np.abs(got[k] - want[k])) <= 1e-4, k
    assert np.max(np.abs(got["depth_fine"] - want["depth_fine"])) <= 8e-4


# --------------------------------------------------------------------------- next rows
def test_generate_rays_matches_oracle():
    """On-device pin-hole ray generation (datasets/ray_utils.py:6-53) vs the oracle restatement."""
    import ctypes
    M = _M()
    H, W = 37, 53
    pose = O.look_at_pose(eye=(1.0, -3.0, 2.0))
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    want_o, want_d = O.get_rays(O.get_ray_directions(H, W, focal), pose)
    rays = torch.empty(H * W, 8, device=DEV)
    c2w = (ctypes.c_float * 12)(*pose.reshape(-1).tolist())
    M._lib.check(M._lib.lib().mnrf_generate_rays(H, W, float(focal), c2w, 0.05, 8.0, M._lib.ptr(rays), M._lib.stream()),
                 "mnrf_generate_rays")
    r = rays.cpu().numpy()
    assert np.max(np.abs(r[:, 0:3] - want_o)) <= 1e-6 and np.max(np.abs(r[:, 3:6] - want_d)) <= 1e-6
    assert np.all(r[:, 6] == np.float32(0.05)) and np.all(r[:, 7] == 8)


@pytest.mark.parametrize("name", ["g12_rays_37x53", "g12_rays_64x64"])
def test_generate_rays_golden(name):
    """G12: mnrf_generate_rays against rays captured from the reference's datasets/ray_utils.py:6-53 + blender.py:159-168."""
    from mirror_nerf_amd import synthetic as SY
    fx = FX.Fixture(name)
    m = fx.meta
    angle = 2 * np.arctan(0.5 * m["W"] / m["focal"])
    rays = SY.device_rays(m["H"], m["W"], DEV, fx.inputs["pose"], m["near"], m["far"], camera_angle_x=angle).cpu().numpy()
    want = fx.outputs["rays"]
    assert rays.shape == want.shape
    assert np.max(np.abs(rays[:, :6] - want[:, :6])) <= 1e-6
    assert np.array_equal(rays[:, 6:], want[:, 6:])


def test_psnr_against_oracle_render():
    """BASELINE north star: PSNR within 0.1 dB of the reference.  The HIP render of a 24x24 view is
    compared with the oracle's render of the same view: the two images differ by ~1e-6, i.e. their
    mutual PSNR is > 90 dB, so any PSNR against a third image agrees to far better than 0.1 dB."""
    from tests.golden import weights as GW
    sds = [GW.apply_tweaks(sd, GW.STRADDLE) for sd in GW.make_state_dict(0, 2)]
    models = {"coarse": _module(sds[0]), "fine": _module(sds[1])}
    rays_np = O.synthetic_rays(24, 24)
    args = dict(predict_normal=True, only_one_field=False, only_one_field_fine_epoch=2, max_recursive_level=1)
    got = _M().batched_inference(models, _emb(), torch.from_numpy(rays_np).to(DEV), 64, 64, False, 32768, args=args,
                                 trace_secondary_rays=True)
    want = O.render_eval({"coarse": sds[0], "fine": sds[1]}, EMB_O, rays_np, 64, 64, False, 32768, args)
    assert O.psnr(got["rgb_fine"].numpy(), want["rgb_fine"]) > 90.0
    target = np.random.RandomState(0).uniform(size=want["rgb_fine"].shape).astype(np.float32)
    assert abs(O.psnr(got["rgb_fine"].numpy(), target) - O.psnr(want["rgb_fine"], target)) < 1e-3
    # keep the frame on the device and score it there; hand only the per-ray maps to the host
    from mirror_nerf_amd import metrics
    dev_out = _M().batched_inference(models, _emb(), torch.from_numpy(rays_np).to(DEV), 64, 64, False, 32768, args=args,
                                     trace_secondary_rays=True, to_cpu=False)
    assert abs(float(metrics.psnr(dev_out["rgb_fine"], torch.from_numpy(target).to(DEV))) - O.psnr(want["rgb_fine"], target)) < 1e-3
    maps = _M().batched_inference(models, _emb(), torch.from_numpy(rays_np).to(DEV), 64, 64, False, 32768, args=args,
                                  trace_secondary_rays=True, to_cpu="maps")
    assert all(not v.is_cuda and (v.dim() == 1 or v.shape[1] <= 3) for v in maps.values())
    assert {"rgb_fine", "depth_fine", "mirror_mask_fine", "opacity_fine"} <= set(maps)
    assert not any(k.startswith(("weights_", "z_vals_", "pred_normal_")) for k in maps)
    for k, v in maps.items():
        assert torch.equal(v, got[k]), k


def test_training_reduces_loss():
    """Twenty Adam steps on a fixed synthetic batch through the HIP forward/backward kernels."""
    from mirror_nerf_amd import training
    torch.manual_seed(0)
    system = _M().NeRFSystem(training.default_hparams(perturb=0.0, noise_std=0.0)).to(DEV)
    with torch.no_grad():
        for m in (system.nerf_coarse, system.nerf_fine):
            m.sigma.weight.mul_(20.0)
            m.sigma.bias.fill_(1.0)
    opt = torch.optim.Adam(list(system.parameters()), lr=5e-4)
    rays = torch.from_numpy(O.synthetic_rays(16, 16)).to(DEV)
    g = torch.Generator(device=DEV)
    g.manual_seed(0)
    target = torch.rand(256, 3, device=DEV, generator=g)
    gt = (torch.rand(256, device=DEV, generator=g) < 0.25).float()
    losses = [float(training.train_step(system, opt, rays, target, gt).item()) for _ in range(20)]
    # random per-ray targets leave a large irreducible error: require a steady decrease, not a collapse
    assert all(np.isfinite(losses)) and losses[-1] < 0.97 * losses[0] and losses[10] < losses[0], losses


def test_fewer_encoding_bands_on_the_kernels_for_10_and_4(precision):
    """--N_emb_xyz 6 --N_emb_dir 2 (opt.py:35-46): MirrorNeRF(39, 15) + Embedding(6) / Embedding(2).  The kernels always evaluate
    10 / 4 bands; the weight columns of the absent bands are packed as zeros (weights.canonical).  `MirrorNeRF.forward` on
    (B, 3 + 15) inputs and `render_rays` against the oracle evaluated with 6 / 2 bands (pinned by fixture G16, whose forward
    dict and gradients the HIP path is held to in tests/test_hip_backward.py); the default-size guards still raise."""
    M = _M()
    fx = FX.Fixture("g16_nemb_6_2_train_grads")
    sd = fx.state_dicts()[1]
    m = M.MirrorNeRF(in_channels_xyz=39, in_channels_dir=15, predict_normal=True, predict_mirror_mask=True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(DEV)
    emb = {"xyz": M.Embedding(6), "dir": M.Embedding(2)}
    rs = np.random.RandomState(5)
    xyz = rs.uniform(-3, 3, (777, 3)).astype(np.float32)
    d = rs.normal(size=(777, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    x = torch.cat([torch.from_numpy(xyz).to(DEV), emb["dir"](torch.from_numpy(d).to(DEV))], 1)
    assert x.shape[1] == 18
    with torch.no_grad():
        got = _np(m(x, compute_normal=True, sigma_only=False, embedding_xyz=emb["xyz"], embedding_dir=emb["dir"]))
        sonly = _np(m(x[:, :3].contiguous(), compute_normal=False, sigma_only=True, embedding_xyz=emb["xyz"], embedding_dir=emb["dir"]))
    want = O.field_forward(sd, x.cpu().numpy(), False, True, n_freqs_xyz=6)
    for k in ("sigma", "rgb", "pred_normal", "is_mirror"):
        scale = max(1.0, float(np.abs(want[k]).max()))
        assert float(np.max(np.abs(got[k].reshape(want[k].shape) - want[k]))) <= 1e-4 * scale, k
    assert float(np.max(np.abs(sonly["sigma"].reshape(-1) - want["sigma"].reshape(-1)))) <= 1e-4 * max(1.0, float(np.abs(want["sigma"]).max()))
    with pytest.raises(NotImplementedError):
        m(x, embedding_xyz=M.Embedding(10), embedding_dir=emb["dir"])
    rays = torch.from_numpy(fx.inputs["rays"]).to(DEV)
    models = {"coarse": m, "fine": m}
    with torch.no_grad():
        res = _np(M.render_rays(models, emb, rays, 64, False, 0, 0, 64, 32768, False, True, compute_normal=False))
    orc = O.render_rays({"coarse": sd, "fine": sd}, {"xyz": 6, "dir": 2}, fx.inputs["rays"], 64, False, 0, 0, 64, 32768, False, True,
                        compute_normal=False)
    for k in ("rgb_fine", "depth_fine", "opacity_fine", "mirror_mask_fine"):
        assert float(np.max(np.abs(res[k] - orc[k]))) <= (8e-4 if k.startswith("depth") else 1e-4), k
    with pytest.raises(NotImplementedError):
        M.render_rays(models, _emb(), rays, 64, False, 0, 0, 64, 32768, False, True, compute_normal=False)


# --------------------------------------------------------------------------- split-f16 vs fp32 at full size
def test_split_agrees_with_fp32_chain_full_chunk(precision):
    """BASELINE config 2 chunk (32768 rays x 192 samples = 6.29 M evaluations): the split-f16 kernel against the
    bit-exact fp32 kernel on the same samples.  Tolerance 2e-5 abs on sigma (measured 3e-6), 1e-5 on the
    squashed outputs (measured 1e-7) -- an order of magnitude inside the 1e-4 of the north star."""
    if precision != "split":
        pytest.skip("one comparison covers both")
    from mirror_nerf_amd import mirror_nerf as MN
    from tests.golden import weights as GW
    sd = GW.apply_tweaks(GW.make_state_dict(0, 2)[1], GW.STRADDLE)
    m = _module(sd)
    rays = torch.from_numpy(O.synthetic_rays(800, 800)[300 * 800: 300 * 800 + 32768]).to(DEV)
    z = torch.linspace(0.05, 8.0, 192, device=DEV).repeat(32768, 1).contiguous()
    de = _emb()["dir"](rays[:, 3:6].contiguous())
    B = 32768 * 192
    res = {}
    for mode in ("fp32", "split"):
        MN.set_precision(mode)
        res[mode] = MN.field_forward(m, B, rays=rays, z_vals=z, spr=192, dir_emb=de, dir_stride=27)
    for k, tol in (("sigma", 2e-5), ("rgb", 1e-5), ("is_mirror", 1e-5), ("pred_normal", 2e-5)):
        a, b = res["fp32"][k], res["split"][k]
        assert bool(torch.isfinite(b).all()), k
        scale = max(1.0, float(a.abs().max())) if k == "sigma" else 1.0
        err = float((a - b).abs().max())
        assert err <= tol * scale, f"{k}: split vs fp32 {err:.3e} > {tol * scale:.1e}"


def test_dynamic_tile_queue_of_the_48_sample_kernels(precision):
    """Round 3: launches of the 48-samples-per-wave kernels with more tiles than CUs run one resident workgroup per CU that
    takes its tiles from a global counter (FieldArgs::tile_queue; the last workgroup out resets the counter pair).  A tile
    handed out twice, skipped, or a counter left dirty for the next launch would show as samples that differ from the
    32-samples-per-wave kernels (static grid; launches that ask for geo_feat take those): ragged sizes around multiples of
    192 x 256, the same launch repeated five times (the ring hands every launch another counter pair; all must be clean), two
    launches in flight on two streams, sigma-only and full."""
    if precision != "split":
        pytest.skip("the queue belongs to the split-f16 kernels")
    from mirror_nerf_amd import mirror_nerf as MN
    from tests.golden import weights as GW
    m = _module(GW.apply_tweaks(GW.make_state_dict(0, 2)[1], GW.STRADDLE))
    e = _emb()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    for B in (192 * cus + 1, 192 * cus * 3 - 7, 192 * (2 * cus + 5)):
        torch.manual_seed(B)
        xyz = (torch.rand(B, 3, device=DEV) * 6 - 3).contiguous()
        de = e["dir"](torch.nn.functional.normalize(torch.randn(B, 3, device=DEV), dim=1))
        want = MN.field_forward(m, B, xyz=xyz, dir_emb=de, dir_stride=27, want_geo=True)      # 32-sample kernels, static grid
        KEYS = ("sigma", "rgb", "is_mirror", "pred_normal")

        def same(got, what):
            for k in KEYS if "rgb" in got else ("sigma",):
                scale = max(1.0, float(want[k].abs().max()))
                assert float((got[k] - want[k]).abs().max()) <= 1e-6 * scale, (B, what, k)
        first = MN.field_forward(m, B, xyz=xyz, dir_emb=de, dir_stride=27)
        same(first, "first launch")
        for rep in range(5):
            got = MN.field_forward(m, B, xyz=xyz, dir_emb=de, dir_stride=27)
            for k in KEYS:
                assert torch.equal(got[k], first[k]), (B, rep, k)
        same(MN.field_forward(m, B, xyz=xyz, sigma_only=True), "sigma-only")
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        s1.wait_stream(torch.cuda.current_stream())
        s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s1):
            a = MN.field_forward(m, B, xyz=xyz, dir_emb=de, dir_stride=27)
        with torch.cuda.stream(s2):
            b = MN.field_forward(m, B, xyz=xyz, dir_emb=de, dir_stride=27)
        torch.cuda.synchronize()
        for k in KEYS:
            assert torch.equal(a[k], first[k]) and torch.equal(b[k], first[k]), (B, "two streams", k)


def test_split_kernel_random_shapes_against_fp32(precision):
    """Race / schedule check of the split kernels (static LDS-DMA schedule, counted waits, in-place operand overwrite):
    eight random sizes and sample counts, forward-only and with the density-gradient pass, against the bit-exact
    kernel.  Bound 5e-5 relative (measured 2e-6); a stale LDS slot or a half-converted operand would be O(1)."""
    if precision != "split":
        pytest.skip("one comparison covers both")
    from mirror_nerf_amd import mirror_nerf as MN
    from tests.golden import weights as GW
    m = _module(GW.apply_tweaks(GW.make_state_dict(0, 2)[1], GW.STRADDLE))
    e = _emb()
    for it in range(8):
        torch.manual_seed(100 + it)
        n = int(torch.randint(500, 20000, (1,)))
        S = [64, 192, 70, 33][it % 4]
        rays = torch.randn(n, 8, device=DEV)
        rays[:, 3:6] = torch.nn.functional.normalize(rays[:, 3:6], dim=1)
        z = torch.sort(torch.rand(n, S, device=DEV) * 6 + 0.1, 1)[0].contiguous()
        de = e["dir"](rays[:, 3:6].contiguous())
        out = {}
        for mode in ("fp32", "split"):
            MN.set_precision(mode)
            out[mode] = MN.field_forward(m, n * S, rays=rays, z_vals=z, spr=S, dir_emb=de, dir_stride=27,
                                         grad_normal=(it % 2 == 0), sigma_only=(it % 4 == 3))
        for k in out["fp32"]:
            if k == "normal":
                continue        # noise-dominated per sample (tests/golden/fixtures.py)
            a, b = out["fp32"][k], out["split"][k]
            d = float((a - b).abs().max() / max(1.0, float(a.abs().max())))
            assert d < 5e-5, (it, k, d)


def test_verify_split_reports_range_problems(precision):
    """mirror_nerf.verify_split: ~1e-6 on a normal model; a model whose activations leave the range of the f16 hi/lo
    pairs (weights of the first layer scaled until h1 ~ 1e6) is reported as such -- the documented limit of the split."""
    if precision != "split":
        pytest.skip("one check covers both")
    from mirror_nerf_amd import mirror_nerf as MN
    from tests.golden import weights as GW
    sd = GW.make_state_dict(0, 1)[0]
    m = _module(sd)
    rep = MN.verify_split(m)
    assert max(rep.values()) < 2e-5, rep
    big = {k: v.copy() for k, v in sd.items()}
    big["xyz_encoding_1.0.weight"] *= 3e6
    rep = MN.verify_split(_module(big))
    assert max(rep.values()) > 1e-3, rep
    assert MN.PRECISION == "split"


@pytest.mark.parametrize("switch", ["MNRF_SPLIT32=1", "MNRF_SPLIT48=0"])
def test_split32_tuning_passes_the_parity_suite(switch):
    """The forward-only split launches run on the 48-samples-per-wave tuning by default (csrc/mnrf_field_split3.hip, DESIGN
    9.2).  MNRF_SPLIT48=0 puts them back on the 32-samples-per-wave kernels (still the ones a geo_feat request gets),
    MNRF_SPLIT32=1 on the 32x32x16 tuning (csrc/mnrf_field_split32.inc: fewer cycles, lower clock -- DESIGN 9.1).  The
    switches are read once per process, so the field / render / recursion parity tests, the random-shape schedule check and
    the range-guard tests run again in a child process for each."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    name, val = switch.split("=")
    env = dict(os.environ, **{name: val})
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.join(root, "tests", "test_hip_parity.py"),
                        os.path.join(root, "tests", "test_hip_guard.py"), "-k",
                        "split and not split32 or field_golden or render_rays_golden or recursion_eval or trained_weights_eval "
                        "or ragged or threshold or saturation or falls_back"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]


# --------------------------------------------------------------------------- ray-fused eval pass (maps only)
MAP_KEYS = ("rgb_fine", "depth_fine", "opacity_fine", "mirror_mask_fine", "surface_normal_fine", "x_surface_fine")


@pytest.mark.parametrize("name", ["g4_fine_test", "g11_trained_render_test", "g11_rough_render_test"])
def test_fused_fine_pass_gives_the_same_maps_bit_for_bit(name, precision):
    """`_maps_only` (eval): field evaluation + compositing of the 192-sample fine pass in ONE kernel, head outputs in LDS
    (mnrf_field_composite_fused).  The per-ray maps must equal the two-kernel path's bit for bit (the compositing is the same
    code), meet the fixture like it does, and no per-sample key of the final pass may be produced.  Under the fp32
    arithmetic the request falls back to the two-kernel path (same keys as maps-only would give are still present)."""
    fx = FX.Fixture(name)
    m = fx.meta
    sds = fx.state_dicts()
    models = {"coarse": _module(sds[0]), "fine": _module(sds[1])}
    rays = torch.from_numpy(fx.inputs["rays"]).to(DEV)

    def render(r, **kw):
        with torch.no_grad():      # (with autograd on, render_rays takes the training forward: another kernel family)
            return _np(_M().render_rays(models, _emb(), r, m["N_samples"], m["use_disp"], m["perturb"], m["noise_std"], m["N_importance"],
                                        m["chunk"], m["white_back"], m["test_time"], **kw, **m["kwargs"]))
    got = render(rays)
    fused = render(rays, _maps_only=True)
    for k in MAP_KEYS:
        assert np.array_equal(fused[k], got[k]), k
    fusable = precision == "split" and os.environ.get("MNRF_SPLIT32", "0") != "1" and os.environ.get("MNRF_SPLIT48", "1") != "0"
    if fusable:      # (other tunings of the field kernel: the request falls back to the two-kernel path, keys and all)
        assert "weights_fine" not in fused and "pred_normal_fine" not in fused
    assert "weights_coarse" in fused and "z_vals_fine" in fused
    if name.startswith("g11_"):
        _cmp_trained(name, fused, fx, list(FX.PER_SAMPLE_FINE) + ["normal_coarse", "normal_fine", "pred_normal_coarse"])
    else:
        _cmp(name, fused, {k: v for k, v in fx.outputs.items() if k in fused}, m)
    # ragged ray counts and an empty batch
    for n in (1, 3, 0):
        r = render(rays[:n].contiguous(), _maps_only=True)
        assert r["rgb_fine"].shape == (n, 3)
        if n:
            assert np.array_equal(r["rgb_fine"], got["rgb_fine"][:n])


def test_fused_eval_through_the_recursion(precision):
    """batched_inference with maps_only=True / to_cpu="maps": the same per-ray maps as the full-dict frame, two bounces
    (level-1 reflections are compacted: ragged ray counts go through the fused kernel too)."""
    from tests.golden import weights as GW
    sds = GW.make_state_dict(0, 2)
    for sd in sds:
        GW.apply_tweaks(sd, GW.STRADDLE)
    models = {"coarse": _module(sds[0]), "fine": _module(sds[1])}
    rays = torch.from_numpy(O.synthetic_rays(40, 40)[::3][:500].copy()).to(DEV)
    args = dict(predict_normal=True, only_one_field=False, only_one_field_fine_epoch=2, max_recursive_level=2)
    M = _M()
    full = M.batched_inference(models, _emb(), rays, 64, 128, False, 200, args=args, trace_secondary_rays=True, to_cpu=False)
    maps = M.batched_inference(models, _emb(), rays, 64, 128, False, 200, args=args, trace_secondary_rays=True, to_cpu=False, maps_only=True)
    host = M.batched_inference(models, _emb(), rays, 64, 128, False, 200, args=args, trace_secondary_rays=True, to_cpu="maps")
    assert int((full["mirror_mask_fine"] != 0).sum()) > 0
    for k in MAP_KEYS + ("rgb_fine_reflect", "depth_fine_reflect", "reflect_direction"):
        assert torch.equal(maps[k], full[k]), k
        assert torch.equal(host[k], full[k].cpu()), k
    assert all(v.dim() <= 2 for v in maps.values())
    # the host maps travel through pinned staging buffers that are RE-USED between frames: what a call returns must not
    # alias them (a second frame -- other rays, fewer of them -- must leave the first frame's tensors untouched)
    keep = {k: v.clone() for k, v in host.items()}
    other = M.batched_inference(models, _emb(), rays[100:400].flip(0).contiguous(), 64, 128, False, 200, args=args,
                                trace_secondary_rays=True, to_cpu="maps")
    assert other["rgb_fine"].shape[0] == 300 and not other["rgb_fine"].is_pinned()
    for k, v in host.items():
        assert torch.equal(v, keep[k]), k
    if precision == "split" and os.environ.get("MNRF_SPLIT32", "0") != "1" and os.environ.get("MNRF_SPLIT48", "1") != "0":
        assert "weights_fine" not in maps
    # N_importance = 64 (128 samples per ray) is not the fused launch class: the request is honoured by the two-kernel path
    a = M.batched_inference(models, _emb(), rays, 64, 64, False, 200, args=args, trace_secondary_rays=True, to_cpu=False)
    b = M.batched_inference(models, _emb(), rays, 64, 64, False, 200, args=args, trace_secondary_rays=True, to_cpu=False, maps_only=True)
    assert torch.equal(a["rgb_fine"], b["rgb_fine"]) and torch.equal(a["depth_fine"], b["depth_fine"])
