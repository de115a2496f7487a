"""The oracle against the golden vectors captured from the reference (CPU)."""
import numpy as np
import pytest

from oracle import mirror_nerf_oracle as O
from tests.golden import fixtures as FX

EMB = {"xyz": 10, "dir": 4}


def _cmp(fx, got, skip=(), base=2e-6):
    for k, want in fx.outputs.items():
        if k in skip:
            continue
        assert k in got, f"{fx.name}: missing {k}"
        assert got[k].shape == want.shape, (fx.name, k)
        tol = base
        if k in FX.GRAD_NORMAL_KEYS:
            tol = FX.tolerance(k, fx.meta)
        elif k in ("z_vals_coarse", "pred_normal_coarse") and fx.meta.get("kwargs", {}).get("only_one_field"):
            tol = FX.tolerance(k, fx.meta)  # "coarse" keys hold fine-position samples there
        elif k.startswith(("depth", "x_surface")):
            tol = 8 * base
        err = float(np.max(np.abs(got[k].astype(np.float64) - want))) if want.size else 0.0
        assert err <= tol, f"{fx.name}:{k} max-abs {err:.3e} > {tol:.1e}"
    assert set(got) >= set(fx.outputs)


def test_linspace_and_embedding():
    fx = FX.Fixture("g1_embedding")
    for n in (2, 5, 64, 128, 192):
        assert np.array_equal(O.torch_linspace(0, 1, n), fx.outputs[f"linspace_{n}"])
    assert np.max(np.abs(O.embedding(fx.inputs["x"], 10) - fx.outputs["e10"])) <= 2e-7
    assert np.max(np.abs(O.embedding(fx.inputs["d"], 4) - fx.outputs["e4"])) <= 2e-7
    assert np.array_equal(O.embedding(fx.inputs["x"], 0), fx.outputs["e0"])


def test_field():
    fx = FX.Fixture("g2_field")
    sd = fx.state_dicts()[0]
    x30 = fx.inputs["x30"]
    full = O.field_forward(sd, x30, False, True)
    sonly = O.field_forward(sd, x30[:, :3], True, False)
    for k, want in fx.outputs.items():
        mode, key = k.split("__")
        got = (full if mode == "full" else sonly)[key]
        if key == "geo_feat":
            got = got[:, :8]
        assert np.max(np.abs(got - want)) <= 2e-6, k


@pytest.mark.parametrize("name", FX.names("g3_") + FX.names("g4_") + FX.names("g5_"))
def test_render_rays(name):
    fx = FX.Fixture(name)
    m = fx.meta
    sds = fx.state_dicts()
    models = {"coarse": sds[0]}
    if m["N_importance"] > 0 and not m["kwargs"].get("only_one_field", False):
        models["fine"] = sds[1]
    kw = dict(m["kwargs"])
    for k in ("perturb_rand", "noise_coarse", "noise_fine", "u"):
        if k in fx.inputs:
            kw["_" + k] = fx.inputs[k]
    got = O.render_rays(models, EMB, fx.inputs["rays"], m["N_samples"], m["use_disp"], m["perturb"],
                        m["noise_std"], m["N_importance"], m["chunk"], m["white_back"], m["test_time"], **kw)
    loose = name == "g4_fine_rawinit"  # ill-conditioned on purpose: floor-aware tolerances
    if loose:
        for k, want in fx.outputs.items():
            if k in FX.PER_SAMPLE_FINE:
                continue
            err = float(np.max(np.abs(got[k] - want)))
            assert err <= FX.tolerance(k, m), (k, err)
    else:
        _cmp(fx, got, skip=FX.PER_SAMPLE_FINE)
    # order-insensitive check of the per-sample fine tensors
    if "weights_fine" in fx.outputs and not loose:
        assert np.max(np.abs(got["weights_fine"].sum(1) - fx.outputs["weights_fine"].sum(1))) <= 2e-6
        assert np.all(np.diff(got["z_vals_fine"], axis=1) >= 0)


@pytest.mark.parametrize("name", FX.names("g6_"))
def test_recursion_train(name):
    fx = FX.Fixture(name)
    sds = fx.state_dicts()
    got = O.render_train({"coarse": sds[0], "fine": sds[1]}, EMB, fx.inputs["rays"], fx.meta["hp"],
                         {"mirror_mask": fx.inputs["gt_mask"].copy(), "is_eval": fx.meta["is_eval"],
                          "train_geometry_stage": False})
    assert set(got) - set(("pred_normal_coarse", "pred_normal_fine", "normal_coarse", "normal_fine")) \
        == set(fx.outputs), set(got) ^ set(fx.outputs)
    _cmp(fx, got, skip=FX.PER_SAMPLE_FINE)


@pytest.mark.parametrize("name", FX.names("g7_") + FX.names("g8_") + FX.names("g8b_"))
def test_recursion_eval(name):
    fx = FX.Fixture(name)
    m = fx.meta
    sds = fx.state_dicts()
    noise = [fx.inputs[f"normal_noise_{i}"] for i in range(len([k for k in fx.inputs if k.startswith("normal_noise_")]))]
    got = O.render_eval({"coarse": sds[0], "fine": sds[1]}, EMB, fx.inputs["rays"], m["N_samples"],
                        m["N_importance"], False, m["chunk"], m["args"], normal_noise=iter(noise))
    _cmp(fx, got, skip=FX.PER_SAMPLE_FINE, base=5e-6)


@pytest.mark.parametrize("name", FX.names("g15_c1_"))
def test_config1_through_the_recursion(name):
    """G15: BASELINE config 1 (coarse only, 64 samples, one bounce) through NeRFSystem.forward / batched_inference
    (`select_type = "coarse"`, train.py:147-151, eval.py:132-172), captured by tests/golden/make_golden_config1.py."""
    fx = FX.Fixture(name)
    m = fx.meta
    sd = fx.state_dicts()[0]
    if "_train_" in name:
        got = O.render_train({"coarse": sd}, EMB, fx.inputs["rays"], m["hp"],
                             {"mirror_mask": fx.inputs["gt_mask"].copy(), "is_eval": m["is_eval"], "train_geometry_stage": False})
    else:
        got = O.render_eval({"coarse": sd}, EMB, fx.inputs["rays"], m["N_samples"], 0, False, m["chunk"], m["args"])
    assert not any(k.endswith("_fine") for k in got)
    _cmp(fx, got, base=5e-6)


def test_fewer_encoding_bands_golden():
    """G16 (tests/golden/make_golden_nemb.py): --N_emb_xyz 6 --N_emb_dir 2 through the reference's NeRFSystem.forward; the
    oracle takes the band counts as arguments.  And the identity the HIP path rests on: the same model with its weight
    columns zero-padded to 63 / 27 channels (mirror_nerf_amd.weights.canonical) evaluated with 10 / 4 bands gives the same
    field outputs."""
    fx = FX.Fixture("g16_nemb_6_2_train_grads")
    m = fx.meta
    sds = fx.state_dicts()
    assert sds[0]["xyz_encoding_1.0.weight"].shape == (256, 39) and sds[0]["dir_encoding.0.weight"].shape == (128, 256 + 15)
    emb = {"xyz": m["N_emb_xyz"], "dir": m["N_emb_dir"]}
    got = O.render_train({"coarse": sds[0], "fine": sds[1]}, emb, fx.inputs["rays"], m["hp"],
                         {"mirror_mask": fx.inputs["gt_mask"].copy(), "is_eval": False, "train_geometry_stage": False})
    want = {k: v for k, v in fx.outputs.items() if k != "loss" and not k.startswith("grad__")}
    for k, w in want.items():
        if k in FX.PER_SAMPLE_FINE:
            continue
        err = float(np.max(np.abs(got[k].astype(np.float64) - w))) if w.size else 0.0
        assert err <= FX.tolerance(k, m), (k, err)
    import torch
    from mirror_nerf_amd.weights import canonical, decanonical
    sd = sds[1]
    pad = {k: canonical(k, torch.from_numpy(v)).numpy() for k, v in sd.items()}
    assert pad["xyz_encoding_1.0.weight"].shape == (256, 63) and pad["xyz_encoding_5.0.weight"].shape == (256, 319)
    assert pad["dir_encoding.0.weight"].shape == (128, 283)
    for k in ("xyz_encoding_1.0.weight", "xyz_encoding_5.0.weight", "dir_encoding.0.weight"):
        assert np.array_equal(decanonical(k, torch.from_numpy(pad[k]), sd[k].shape).numpy(), sd[k])
    rs = np.random.RandomState(3)
    xyz = rs.uniform(-3, 3, (200, 3)).astype(np.float32)
    d = rs.normal(size=(200, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    a = O.field_forward(sd, np.concatenate([xyz, O.embedding(d, 2)], 1), False, True, n_freqs_xyz=6)
    b = O.field_forward(pad, np.concatenate([xyz, O.embedding(d, 4)], 1), False, True, n_freqs_xyz=10)
    for k in ("sigma", "rgb", "pred_normal", "is_mirror", "normal"):
        assert np.max(np.abs(a[k] - b[k])) <= 2e-6 * max(1.0, float(np.abs(a[k]).max())), k


@pytest.mark.parametrize("name", ["g12_rays_37x53", "g12_rays_64x64"])
def test_ray_generation_golden(name):
    """G12: the oracle's pin-hole rays against the reference's (datasets/ray_utils.py:6-53, captured by make_golden_rays.py)."""
    fx = FX.Fixture(name)
    m = fx.meta
    o, d = O.get_rays(O.get_ray_directions(m["H"], m["W"], m["focal"]), fx.inputs["pose"])
    want = fx.outputs["rays"]
    assert np.max(np.abs(o - want[:, 0:3])) <= 1e-6 and np.max(np.abs(d - want[:, 3:6])) <= 1e-6
    assert np.all(want[:, 6] == np.float32(m["near"])) and np.all(want[:, 7] == np.float32(m["far"]))


def test_synthetic_rays_shape_and_norm():
    rays = O.synthetic_rays(20, 30)
    assert rays.shape == (600, 8)
    assert np.allclose(np.linalg.norm(rays[:, 3:6], axis=1), 1, atol=1e-6)
    assert np.all(rays[:, 6] == np.float32(0.05)) and np.all(rays[:, 7] == 8)


# ---- a15: fixtures G17 = the reference's models/mirror_nerf_tcnn.py run unchanged over stand-in encoders
#      (tests/golden/make_golden_tcnn.py): everything downstream of the encoder's interpolation is pinned
def test_g17_grid_offsets():
    """Level offsets of the table against models/gridencoder/grid.py:181-194 (GridEncoder.__init__ run with its backend
    stubbed) at the per_level_scale of mirror_nerf_tcnn.py:38 -- oracle and product."""
    from mirror_nerf_amd.mirror_nerf_tcnn import hashgrid_config
    fx = FX.Fixture("g17_grid_offsets")
    for b in fx.meta["bounds"]:
        want = fx.outputs[f"offsets_b{b:g}"]
        assert np.array_equal(O.hashgrid_config(b)["offsets"], want)
        assert np.array_equal(hashgrid_config(b)["offsets"], want)
        assert hashgrid_config(b)["S"] == O.hashgrid_config(b)["S"]


@pytest.mark.parametrize("name", ["g17_tcnn_field_b1", "g17_tcnn_field_b6"])
def test_g17_tcnn_field(name):
    fx = FX.Fixture(name)
    w = FX.tcnn_weights(fx, "w__")
    x6 = fx.inputs["x6"]
    full = O.tcnn_field_forward(w, x6, w["_cfg"], False, True)
    so = O.tcnn_field_forward(w, x6[:, :3], w["_cfg"], True, False)
    assert set(so) == {"sigma", "geo_feat", "pred_normal"}           # pred_normal also when sigma_only (mirror_nerf_tcnn.py:184-192)
    for k, want in fx.outputs.items():
        mode, key = k.split("__")
        got = (full if mode == "full" else so)[key].reshape(want.shape)
        err = float(np.max(np.abs(got - want)))
        assert err <= (2e-6 if key in ("sigma", "geo_feat", "rgb", "is_mirror") else 5e-6), (k, err)


@pytest.mark.parametrize("name", ["g17_tcnn_render_train", "g17_tcnn_render_test"])
def test_g17_tcnn_render_rays(name):
    fx = FX.Fixture(name)
    ws = {"coarse": FX.tcnn_weights(fx, "coarse__", 0), "fine": FX.tcnn_weights(fx, "fine__", 1)}
    tt = fx.meta["test_time"]
    got = O.render_rays(ws, {"xyz": 0, "dir": 0}, fx.inputs["rays"], 64, False, 0, 0, 64, 32768, False, tt, compute_normal=not tt)
    n = 0
    for k, want in fx.outputs.items():
        if k in FX.PER_SAMPLE_FINE:
            continue
        err = float(np.max(np.abs(got[k].astype(np.float64) - want))) if want.size else 0.0
        assert err <= FX.tolerance(k, fx.meta), (k, err)
        n += 1
    assert n >= (8 if tt else 18)


@pytest.mark.parametrize("name", ["g17_tcnn_train_grads"])
def test_g17_tcnn_train_forward(name):
    fx = FX.Fixture(name)
    ws = {"coarse": FX.tcnn_weights(fx, "coarse__", 0), "fine": FX.tcnn_weights(fx, "fine__", 1)}
    got = O.render_train(ws, {"xyz": 0, "dir": 0}, fx.inputs["rays"], fx.meta["hp"],
                         {"mirror_mask": fx.inputs["gt_mask"].copy(), "is_eval": False, "train_geometry_stage": False})
    n = 0
    for k, want in fx.outputs.items():
        if k == "loss" or k.startswith(("grad__", "table_", "z_fine_l")) or k in FX.PER_SAMPLE_FINE:
            continue
        err = float(np.max(np.abs(got[k].astype(np.float64) - want))) if want.size else 0.0
        assert err <= FX.tolerance(k, fx.meta), (k, err)
        n += 1
    assert n >= 20


@pytest.mark.parametrize("name", ["g17_tcnn_eval_l1", "g17_tcnn_eval_l2"])
def test_g17_tcnn_eval_recursion(name):
    """eval.batched_inference with hash-grid models (predicted mask; level 0 traces the chunk, deeper levels compact)."""
    fx = FX.Fixture(name)
    m = fx.meta
    ws = {"coarse": FX.tcnn_weights(fx, "coarse__", 0), "fine": FX.tcnn_weights(fx, "fine__", 1)}
    got = O.render_eval(ws, {"xyz": 0, "dir": 0}, fx.inputs["rays"], m["N_samples"], m["N_importance"], False, m["chunk"], m["args"])
    n = 0
    for k, want in fx.outputs.items():
        if k in FX.PER_SAMPLE_FINE:
            continue
        err = float(np.max(np.abs(got[k].astype(np.float64) - want))) if want.size else 0.0
        assert err <= FX.tolerance(k, m), (k, err)
        n += 1
    assert n >= 12 and int((fx.outputs["mirror_mask_fine"] > 0.5).sum()) > 10


@pytest.mark.parametrize("variant", ["plain", "detach_normal", "detach_mask", "detach_outside", "second_order"])
def test_g17_torch_restatement_gradients(variant):
    """tests/torch_ref.tcnn_field -- the autograd yardstick of the GPU gradient tests -- against the reference's own
    gradients: which head sees geo_feat.detach() under which flag, the raw sigma, the biased mirror head, the second-order
    term through normal = l2n(-d sigma / dx)."""
    import torch
    from tests import torch_ref as TR
    fx = FX.Fixture("g17_tcnn_field_grads")
    w = FX.tcnn_weights(fx, "w__")
    cfg = w.pop("_cfg")
    wt = {k: torch.from_numpy(v).requires_grad_(True) for k, v in w.items()}
    x = torch.from_numpy(fx.inputs["x6"].copy()).requires_grad_(True)
    cot = {k[5:]: torch.from_numpy(v) for k, v in fx.inputs.items() if k.startswith("cot__")}
    if variant == "second_order":
        sigma, _rgb, _pn, _m, normal = TR.tcnn_field_with_normal(wt, x, cfg)
        loss = (sigma * cot["sigma"]).sum() + (normal * cot["normal"]).sum()
    else:
        inside = torch.from_numpy(fx.inputs["inside"]).bool()
        dm = {"detach_mask": True, "detach_outside": ~inside}.get(variant)
        sigma, rgb, pn, m = TR.tcnn_field(wt, x, cfg, detach_normal=variant == "detach_normal", detach_mirror=dm)
        loss = (sigma * cot["sigma"]).sum() + (rgb * cot["rgb"]).sum() + (pn * cot["pred_normal"]).sum() \
            + (m[:, None] * cot["is_mirror"]).sum()
    loss.backward()
    o = fx.outputs
    assert abs(loss.item() - float(o[f"{variant}__loss"])) <= 1e-5 * abs(float(o[f"{variant}__loss"]))
    for k in FX.TCNN_MLP_NAMES:
        want = o[f"{variant}__grad__{k}"]
        g = wt[k].grad.numpy() if wt[k].grad is not None else np.zeros_like(want)
        assert np.max(np.abs(g - want)) <= 2e-5 * np.abs(want).max() + 1e-9, (k, np.max(np.abs(g - want)), np.abs(want).max())
    lv, val, nnz = FX.table_grad_summary(wt["encoder.embeddings"].grad.numpy(), cfg, o[f"{variant}__table_idx"])
    assert nnz == int(o[f"{variant}__table_nnz"])
    assert np.max(np.abs(val - o[f"{variant}__table_val"])) <= 2e-5 * np.abs(o[f"{variant}__table_val"]).max()
    assert np.max(np.abs(lv - o[f"{variant}__table_levels"])) <= 2e-5 * np.abs(o[f"{variant}__table_levels"]).max()
    gx = o[f"{variant}__grad__x6"]
    assert np.max(np.abs(x.grad.numpy() - gx)) <= 1e-4 * np.abs(gx).max()


# ---- a15 (encoder interpolation: parity unpinned): internal consistency of the hash-grid restatement
def test_hashgrid_restatement_is_self_consistent():
    cfg = O.hashgrid_config(bound=6.0)
    assert int(cfg["offsets"][-1]) == 6616280            # table size at bound 6
    assert np.all(np.diff(cfg["offsets"]) % 8 == 0) and np.diff(cfg["offsets"]).max() == 2 ** 19
    rs = np.random.RandomState(0)
    table = rs.uniform(-0.1, 0.1, (int(cfg["offsets"][-1]), 2)).astype(np.float32)
    x = rs.uniform(0.02, 0.98, (64, 3)).astype(np.float32)
    enc, dydx = O.hashgrid_encode(x, table, cfg, want_grad=True)
    assert enc.shape == (64, 32) and dydx.shape == (64, 32, 3)
    for d in range(3):                                    # analytic d/dx against a finite difference (coarse levels)
        x2 = x.copy()
        x2[:, d] += 1e-4
        num = (O.hashgrid_encode(x2, table, cfg).astype(np.float64) - enc) / 1e-4
        assert np.median(np.abs(num[:, :6] - dydx[:, :6, d])) <= 2e-3
    out = O.hashgrid_encode(np.array([[1.5, 0.5, 0.5]], np.float32), table, cfg)
    assert np.all(out == 0)                               # outside the unit box: zeros (gridencoder.cu:118-147)
    # trilinear interpolation reproduces the stored value at a dense-level vertex
    lv = 0
    scale = np.float32(np.exp2(0.0) * 16 - 1.0)
    v = (np.array([[3, 5, 7]], np.float32) - 0.5) / scale + 1e-7
    res = int(np.ceil(scale)) + 1
    idx = 3 + 5 * (res + 1) + 7 * (res + 1) ** 2
    assert np.allclose(O.hashgrid_encode(v.astype(np.float32), table, cfg)[0, :2], table[idx], atol=1e-5)


def test_sh4_is_orthonormal_on_the_sphere():
    rs = np.random.RandomState(1)
    d = rs.normal(size=(200000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    Y = O.sh4(d).astype(np.float64)
    gram = 4 * np.pi * (Y.T @ Y) / d.shape[0]
    assert np.max(np.abs(gram - np.eye(16))) < 0.03


def test_torch_port_matches_oracle():
    """oracle/torch_port.py (the plain-torch CPU baseline bench.py times) against the pinned numpy oracle: eval
    semantics, one and two bounces, all-mirror and straddling mirror heads."""
    import torch
    from mirror_nerf_amd import synthetic as SY
    from oracle import torch_port as TP
    rays = O.synthetic_rays(40, 40)[::13][:96].copy()
    for tweaks, levels in ((SY.ALL_MIRROR, 1), (SY.STRADDLE, 2)):
        sds = [SY.apply_tweaks(sd, tweaks) for sd in SY.make_state_dict(0, 2)]
        args = dict(predict_normal=True, only_one_field=False, only_one_field_fine_epoch=2, max_recursive_level=levels)
        want = O.render_eval({"coarse": sds[0], "fine": sds[1]}, {"xyz": 10, "dir": 4}, rays, 64, 128, False, 32768, args)
        mt = {k: {n: torch.from_numpy(v) for n, v in sd.items()} for k, sd in zip(("coarse", "fine"), sds)}
        got = TP.render_eval(mt, torch.from_numpy(rays), 64, 128, 32768, max_level=levels)
        for k in ("rgb_fine", "depth_fine", "mirror_mask_fine", "opacity_fine", "surface_normal_fine"):
            assert np.abs(got[k].numpy() - want[k]).max() <= 2e-6, k


def test_torch_port_config1_matches_oracle():
    """oracle/torch_port.render_train_coarse (what bench.py times as BASELINE config 1 on the host cores) against the pinned
    numpy oracle's train-semantics recursion (render_train, fixtures G6 / G15): coarse only, GT mirror mask, compacted reflections."""
    import torch
    from mirror_nerf_amd import synthetic as SY
    from oracle import torch_port as TP
    rays = O.synthetic_rays(40, 40)[::11][:120].copy()
    gt = (np.arange(rays.shape[0]) % 4 == 1).astype(np.float32)
    sd = SY.apply_tweaks(SY.make_state_dict(0, 1)[0], SY.STRADDLE)
    hp = dict(N_samples=64, N_importance=0, use_disp=False, perturb=0, noise_std=0, chunk=32768, trace_secondary_rays=True,
              only_one_field=False, max_recursive_level=1, only_trace_rays_in_mirrors=True, for_vis=False)
    want = O.render_train({"coarse": sd}, EMB, rays, hp, {"mirror_mask": gt.copy(), "is_eval": False, "train_geometry_stage": False})
    got = TP.render_train_coarse({"coarse": {n: torch.from_numpy(v) for n, v in sd.items()}}, torch.from_numpy(rays),
                                 torch.from_numpy(gt), 64, 32768)
    assert np.abs(got.numpy() - want["rgb_coarse"]).max() <= 2e-6


@pytest.mark.parametrize("name", [n for n in FX.names("g13_") if not n.endswith("_eval")])
def test_render_rays_without_optional_heads(name):
    """G13: the oracle on models without the normal / mirror-mask heads (models/mirror_nerf.py:80-99)."""
    from tests.golden import weights as GW
    fx = FX.Fixture(name)
    m = fx.meta
    sds = GW.make_state_dict(m["seed"], 2, predict_normal=m["predict_normal"], predict_mirror_mask=m["predict_mirror_mask"])
    for sd in sds:
        GW.apply_tweaks(sd, m["tweaks"])
    got = O.render_rays({"coarse": sds[0], "fine": sds[1]}, EMB, fx.inputs["rays"], 64, False, 0, 0, 64, 32768, False, m["test_time"],
                        **m["kwargs"])
    for k, want in fx.outputs.items():
        if k in FX.PER_SAMPLE_FINE or k.startswith("normal_"):
            continue
        tol = 2e-2 if k in FX.GRAD_NORMAL_KEYS else 5e-6
        assert np.max(np.abs(got[k].astype(np.float64) - want)) <= tol, (name, k)
    assert set(fx.outputs) - {"pred_normal_fine", "pred_normal_coarse", "normal_fine", "normal_coarse"} <= set(got)


def test_oracle_is_as_close_to_the_fp64_truth_as_the_reference():
    """Fixture G14 (make_golden_truth64.py): the reference in float64 on the inputs of g3_coarse64_train.  On the keys
    derived from the normalised density gradient the reference's own fp32 run is noise-dominated (8e-3 .. 1e-2 composited,
    O(1) on single samples); the oracle has to be as close to the truth as that run is (see the GPU twin of this test in
    tests/test_hip_parity.py)."""
    import os
    fx = FX.Fixture("g3_coarse64_train")
    m = fx.meta
    got = O.render_rays({"coarse": fx.state_dicts()[0]}, EMB, fx.inputs["rays"], m["N_samples"], m["use_disp"], m["perturb"],
                        m["noise_std"], m["N_importance"], m["chunk"], m["white_back"], m["test_time"], **m["kwargs"])
    z = np.load(os.path.join(os.path.dirname(FX.__file__), "g14_truth64_g3_coarse64_train.npz"))
    n = 0
    for f in z.files:
        if not f.startswith("out64__"):
            continue
        k = f[7:]
        truth = z[f]
        e_ref = np.abs(fx.outputs[k].astype(np.float64) - truth)
        e_orc = np.abs(got[k].astype(np.float64) - truth)
        scale = 8.0 if k.startswith(("depth", "x_surface", "z_vals")) else 1.0
        if k == "normal_coarse":
            assert e_orc.mean() <= 2.0 * e_ref.mean() + 1e-6, (k, e_orc.mean(), e_ref.mean())
        else:
            assert e_orc.max() <= 2.0 * e_ref.max() + 2e-6 * scale, (k, e_orc.max(), e_ref.max())
        n += 1
    assert n >= 10


@pytest.mark.parametrize("base", ["g4_fine_train", "g11_trained_render_train"])
def test_oracle_fine_pass_is_as_close_to_the_fp64_truth_as_the_reference(base):
    """Fixtures G14 of the FINE pass (make_golden_truth64.truth_fine): the reference in float64 at the fine depths its
    fp32 run drew, on random-init and on trained weights.  Given those depths (`_z_fine`) the oracle has to be as close to
    that truth as the reference's own fp32 run: this is what justifies the 2e-2 / 4 x floor allowances of the plain
    fixture comparison on the fine pass and on trained weights, not only on the coarse random-init case."""
    import os
    fx = FX.Fixture(base)
    m = fx.meta
    sds = fx.state_dicts()
    got = O.render_rays({"coarse": sds[0], "fine": sds[1]}, EMB, fx.inputs["rays"], m["N_samples"], m["use_disp"], m["perturb"],
                        m["noise_std"], m["N_importance"], m["chunk"], m["white_back"], m["test_time"],
                        _z_fine=fx.outputs["z_vals_fine"], **m["kwargs"])
    z = np.load(os.path.join(os.path.dirname(FX.__file__), f"g14_truth64_{base}.npz"))
    n = 0
    for f in z.files:
        if not f.startswith("out64__"):
            continue
        k = f[7:]
        truth = z[f]
        e_ref = np.abs(fx.outputs[k].astype(np.float64) - truth)
        e_orc = np.abs(got[k].astype(np.float64) - truth)
        scale = 8.0 if k.startswith(("depth", "x_surface", "z_vals")) else 1.0
        if k == "normal_fine":       # per sample: vanishing gradients carry O(1) errors in any fp32 run
            assert e_orc.mean() <= 2.0 * e_ref.mean() + 1e-6, (k, e_orc.mean(), e_ref.mean())
        else:
            assert e_orc.max() <= 2.0 * e_ref.max() + 2e-6 * scale, (k, e_orc.max(), e_ref.max())
        n += 1
    assert n >= 9


@pytest.mark.parametrize("base", ["g6_train_gt_compact", "g6_train_pred_straddle"])
def test_oracle_recursion_primary_level_is_as_close_to_the_fp64_truth_as_the_reference(base):
    """Fixtures G14 for two RECURSION fixtures (make_golden_truth64.truth_recursion_level0): the noise-dominated keys of G6
    (surface_normal_grad_*, normal_dif_*) belong to the primary render -- the recursion only re-blends rgb_* -- so their truth is the
    reference's render_rays in float64 on the fixture's rays and weights, at the fine depths its fp32 run drew.  The oracle's render
    of the same call has to be as close to it as the values NeRFSystem.forward left in the fixture."""
    import os
    fx = FX.Fixture(base)
    hp = fx.meta["hp"]
    sds = fx.state_dicts()
    got = O.render_rays({"coarse": sds[0], "fine": sds[1]}, EMB, fx.inputs["rays"], hp["N_samples"], hp["use_disp"], hp["perturb"],
                        hp["noise_std"], hp["N_importance"], hp["chunk"], False, compute_normal=hp["trace_secondary_rays"],
                        _z_fine=fx.outputs["z_vals_fine"])
    z = np.load(os.path.join(os.path.dirname(FX.__file__), f"g14_truth64_{base}.npz"))
    thresholded = fx.meta.get("gt_mode") == "invalid"      # train.py:155-166: the fixture holds the hard predicted masks
    n = 0
    for f in z.files:
        if not f.startswith("out64__"):
            continue
        k = f[7:]
        if thresholded and k.startswith("mirror_mask"):
            continue
        truth = z[f]
        e_ref = np.abs(fx.outputs[k + "_direct" if k in ("rgb_coarse", "rgb_fine") else k].astype(np.float64) - truth)
        e_orc = np.abs(got[k].astype(np.float64) - truth)
        scale = 8.0 if k.startswith(("depth", "x_surface", "z_vals")) else 1.0
        assert e_orc.max() <= 2.0 * e_ref.max() + 2e-6 * scale, (k, e_orc.max(), e_ref.max())
        n += 1
    assert n >= 18


def test_sh4_against_scipy_real_spherical_harmonics():
    """Config 5 is "parity unpinned" (tinycudann and the CUDA SH extension cannot run here, the reference holds no vectors),
    but its view encoding has an independent closed form: the degree-4 real spherical harmonics.  The oracle's restatement of
    models/shencoder/src/shencoder.cu:49-79 -- what the HIP kernels are tested against -- equals scipy's Y_l^|m| combined the
    usual way (sqrt 2 Re for m > 0, sqrt 2 Im for m < 0, Condon-Shortley phase kept as the CUDA polynomials keep it) on all 16
    components: one half of the hash-grid field's input encoding has an outside anchor."""
    from scipy.special import sph_harm_y
    rs = np.random.RandomState(0)
    d = rs.normal(size=(4000, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    theta, phi = np.arccos(np.clip(d[:, 2], -1, 1)), np.arctan2(d[:, 1], d[:, 0])
    cols = []
    for l in range(4):
        for m in range(-l, l + 1):
            Y = sph_harm_y(l, abs(m), theta, phi)
            cols.append(Y.real if m == 0 else (np.sqrt(2) * Y.real if m > 0 else np.sqrt(2) * Y.imag))
    want = np.stack(cols, 1)
    got = O.sh4(d.astype(np.float32)).astype(np.float64)
    assert got.shape == (4000, 16)
    assert np.max(np.abs(got - want)) <= 5e-7
