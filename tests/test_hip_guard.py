"""GPU tests of the range guard of the split-f16 arithmetic (include/mnrf.h MNRF_GUARD_*, mirror_nerf.check_guard):
the kernels raise a sticky device flag when an operand leaves the range the hi/lo f16 pairs carry at fp32 accuracy, the
drivers read it at their sync points, pin the model to the exact fp32 kernels and repeat the work -- a result computed out
of range is never returned."""
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ARGS = dict(predict_normal=True, only_one_field=False, only_one_field_fine_epoch=2, max_recursive_level=1)


@pytest.fixture(autouse=True)
def split_precision():
    from mirror_nerf_amd import mirror_nerf as MN
    old = MN.PRECISION
    MN.set_precision("split")
    yield
    MN.set_precision(old)


def _models(tweaks):
    from mirror_nerf_amd import synthetic as SY
    models, _ = SY.build_models(DEV, tweaks, seed=0)
    return models


def _emb():
    import mirror_nerf_amd as M
    return {"xyz": M.Embedding(10), "dir": M.Embedding(4)}


def _rays(n=96):
    from mirror_nerf_amd import synthetic as SY
    return SY.device_rays(40, 40, DEV)[::13][:n].contiguous()


def _words(models):
    from mirror_nerf_amd import mirror_nerf as MN
    return MN.guard_words(list(models.values()))


def _fp32_render(tweaks, rays, **kw):
    """The same render on a fresh pair pinned to the exact kernels."""
    import mirror_nerf_amd as M
    from mirror_nerf_amd import mirror_nerf as MN
    models = _models(tweaks)
    MN.set_precision("fp32")
    try:
        return M.render_rays(models, _emb(), rays, 64, False, 0, 0, 64, test_time=True, compute_normal=False, **kw)
    finally:
        MN.set_precision("split")


def test_random_init_stays_clean():
    import mirror_nerf_amd as M
    from mirror_nerf_amd import mirror_nerf as MN
    from mirror_nerf_amd import synthetic as SY
    models = _models(SY.OPAQUE)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        M.render_rays(models, _emb(), _rays(), 64, False, 0, 0, 64, test_time=True, compute_normal=False)
        M.batched_inference(models, _emb(), _rays(), 64, 64, False, 32768, args=ARGS, trace_secondary_rays=True, to_cpu=False)
    assert _words(models) == [0, 0]
    assert all(MN.precision_of(m) == "split" for m in models.values())


@pytest.mark.parametrize("level,trips", [(60000.0, False), (70000.0, True)])
def test_saturation_threshold_is_the_f16_maximum(level, trips):
    """A constant first-layer output of `level`: 60 000 is carried exactly by the hi/lo pair, 70 000 saturates hi."""
    import mirror_nerf_amd as M
    from mirror_nerf_amd import mirror_nerf as MN
    m = _models([])["coarse"]
    with torch.no_grad():
        m.xyz_encoding_1[0].weight.zero_()
        m.xyz_encoding_1[0].bias.fill_(level)
        m.xyz_encoding_2[0].weight.mul_(1e-6)     # keep what follows small
    x = torch.rand(300, 3, device=DEV)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        out = m(x, compute_normal=False, sigma_only=True, embedding_xyz=M.Embedding(10))
    assert (MN.precision_of(m) == "fp32") == trips
    assert bool(rec) == trips
    if trips:
        assert "f16 maximum" in str(rec[0].message)
        # what was returned is the fp32 evaluation
        MN.set_precision("fp32")
        m2 = _models([])["coarse"]
        m2.load_state_dict(m.state_dict())
        want = m2.to(DEV)(x, compute_normal=False, sigma_only=True, embedding_xyz=M.Embedding(10))
        assert torch.equal(out["sigma"], want["sigma"])


def test_render_rays_falls_back_and_returns_the_fp32_result():
    import mirror_nerf_amd as M
    from mirror_nerf_amd import mirror_nerf as MN
    from mirror_nerf_amd import synthetic as SY
    tweaks = SY.OPAQUE + [["xyz_encoding_3.0.weight", "mul", 1e7]]       # activations of ~1e6 from layer 3 on
    models = _models(tweaks)
    rays = _rays()
    with pytest.warns(RuntimeWarning, match="split-f16 arithmetic left its range"):
        got = M.render_rays(models, _emb(), rays, 64, False, 0, 0, 64, test_time=True, compute_normal=False)
    assert all(MN.precision_of(m) == "fp32" for m in models.values())
    want = _fp32_render(tweaks, rays)
    for k in ("rgb_fine", "depth_fine", "mirror_mask_fine", "opacity_fine", "weights_coarse"):
        assert torch.equal(got[k], want[k]), k
    # sticky: later calls run on the exact kernels without another warning; reset_guard undoes the pin
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        M.render_rays(models, _emb(), rays, 64, False, 0, 0, 64, test_time=True, compute_normal=False)
    MN.reset_guard(models["coarse"])
    assert MN.precision_of(models["coarse"]) == "split"


def test_positions_beyond_the_fast_sincos_range_fall_back():
    import mirror_nerf_amd as M
    from mirror_nerf_amd import mirror_nerf as MN
    from mirror_nerf_amd import synthetic as SY
    models = _models(SY.OPAQUE)
    rays = _rays(64).clone()
    rays[:, 0] += 100.0                  # origins at x ~ 100: 2^9 * x > 2^15
    with pytest.warns(RuntimeWarning, match=r"\|x\| >= 64"):
        got = M.render_rays(models, _emb(), rays, 64, False, 0, 0, 64, test_time=True, compute_normal=False)
    assert models["coarse"].__dict__["_mnrf_guard_trips"] == 1
    want = _fp32_render(SY.OPAQUE, rays)
    for k in ("rgb_fine", "depth_fine", "opacity_fine"):
        assert torch.equal(got[k], want[k]), k
    # round 5: a range-only trip is a property of the CALL's rays -- the repeated work ran on fp32, the models are back on split
    assert MN.precision_of(models["fine"]) == "split" and MN.precision_of(models["coarse"]) == "split"
    assert _words(models) == [0, 0]
    near = _rays(64)
    with warnings.catch_warnings():
        warnings.simplefilter("error")           # ... and the next call, with rays in range, runs clean on the split arithmetic
        M.render_rays(models, _emb(), near, 64, False, 0, 0, 64, test_time=True, compute_normal=False)
    # a model that keeps meeting such rays is pinned after RANGE_TRIPS_BEFORE_PIN calls
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(MN.RANGE_TRIPS_BEFORE_PIN):
            M.render_rays(models, _emb(), rays, 64, False, 0, 0, 64, test_time=True, compute_normal=False)
    assert MN.precision_of(models["coarse"]) == "fp32"


def test_far_plane_at_twenty_stays_on_the_split_arithmetic():
    """A scale_factor-style scene (real_arkit captures, BASELINE config 4: run.sh:14-15, 47-48 set near / far per scene) whose far
    plane reaches 20: positions up to |x| ~ 24.  Round 4 tripped the encoding-range bit at |x| >= 16 and pinned the model to fp32
    (3.8x slower) for good; the four-term sin/cos reduction is exact to |x| < 64: no trip, and the split render still matches the
    exact fp32 one at the parity tolerance."""
    import mirror_nerf_amd as M
    from mirror_nerf_amd import mirror_nerf as MN
    from mirror_nerf_amd import synthetic as SY
    models = _models(SY.OPAQUE)
    rays = _rays(256).clone()
    rays[:, 7] = 20.0
    rays[:, 0:3] *= 1.2
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        got = M.render_rays(models, _emb(), rays, 64, False, 0, 0, 64, test_time=True, compute_normal=False)
    assert _words(models) == [0, 0] and MN.precision_of(models["fine"]) == "split"
    want = _fp32_render(SY.OPAQUE, rays)
    assert float((got["z_vals_fine"].max())) > 19.0
    for k in ("rgb_fine", "opacity_fine"):
        assert float((got[k] - want[k]).abs().max()) <= 1e-4, k
    assert float((got["depth_fine"] - want["depth_fine"]).abs().max()) <= 1e-4 * 20.0


def test_non_finite_weight_is_flagged_at_pack_time():
    import mirror_nerf_amd as M
    from mirror_nerf_amd import mirror_nerf as MN
    m = _models([])["coarse"]
    with torch.no_grad():
        m.xyz_encoding_4[0].weight[3, 5] = 1e6
    x = torch.rand(64, 3, device=DEV)
    with pytest.warns(RuntimeWarning, match="weight"):
        m(x, compute_normal=False, sigma_only=True, embedding_xyz=M.Embedding(10))
    assert MN.precision_of(m) == "fp32"


def test_frame_driver_and_training_step_fall_back():
    """batched_inference reads the flag once per frame; training.train_step once per step (after the backward)."""
    import mirror_nerf_amd as M
    from mirror_nerf_amd import mirror_nerf as MN
    from mirror_nerf_amd import synthetic as SY
    from mirror_nerf_amd import training
    tweaks = SY.ALL_MIRROR + [["xyz_encoding_6.0.weight", "mul", 1e8]]
    models = _models(tweaks)
    rays = _rays()
    with pytest.warns(RuntimeWarning):
        got = M.batched_inference(models, _emb(), rays, 64, 64, False, 32768, args=ARGS, trace_secondary_rays=True, to_cpu=False)
    assert all(MN.precision_of(m) == "fp32" for m in models.values())
    MN.set_precision("fp32")
    want = M.batched_inference(_models(tweaks), _emb(), rays, 64, 64, False, 32768, args=ARGS, trace_secondary_rays=True, to_cpu=False)
    MN.set_precision("split")
    for k in ("rgb_fine", "depth_fine", "mirror_mask_fine"):
        assert torch.equal(got[k], want[k]), k

    # training: the same step on a pinned-fp32 twin gives the same loss and gradients
    def system():
        torch.manual_seed(0)
        s = M.NeRFSystem(training.default_hparams()).to(DEV)
        with torch.no_grad():
            s.nerf_fine.xyz_encoding_2[0].weight.mul_(1e7)
        return s
    batch = (rays, torch.rand(rays.shape[0], 3, device=DEV), (torch.rand(rays.shape[0], device=DEV) < 0.3).float())
    s2 = system()
    hp = dict(perturb=0.0, noise_std=0.0)
    for k, v in hp.items():
        setattr(s2.hparams, k, v)
    class NoStep:      # leaves the weights alone (SGD with lr = 0 turns an inf gradient into a NaN weight)
        def __init__(self, params):
            self.params = list(params)

        def zero_grad(self, set_to_none=True):
            for q in self.params:
                q.grad = None

        def step(self):
            pass
    s2.nerf_fine.__dict__["_mnrf_precision"] = "fp32"
    l2 = training.train_step(s2, NoStep(s2.parameters()), *batch)
    # MNRF_GUARD_SYNC=1: the tripping step ITSELF is recomputed on the exact kernels before the optimizer sees it
    s4 = system()
    for k, v in hp.items():
        setattr(s4.hparams, k, v)
    old_mode, training.GUARD_MODE = training.GUARD_MODE, "sync"
    try:
        with pytest.warns(RuntimeWarning, match="affected work is repeated"):
            l4 = training.train_step(s4, NoStep(s4.parameters()), *batch)
    finally:
        training.GUARD_MODE = old_mode
    assert float(l4) == float(l2) and MN.precision_of(s4.nerf_fine) == "fp32" and MN.precision_of(s4.nerf_coarse) == "split"      # only the model that tripped
    for (n, p), q in zip(s4.named_parameters(), s2.parameters()):
        if p.grad is not None:
            assert torch.equal(torch.nan_to_num(p.grad), torch.nan_to_num(q.grad)), n
    # The DEFAULT ("skip"): with a fused Adam the tripping step's update is skipped on the device -- weights and optimizer
    # state bit-identical afterwards, no host read in the step --, the next step reports it, runs on fp32 and does update
    assert training.GUARD_MODE == "skip"
    s5 = system()
    for k, v in hp.items():
        setattr(s5.hparams, k, v)
    o5 = torch.optim.Adam(list(s5.parameters()), lr=5e-4, fused=True)
    before = [q.detach().clone() for q in s5.parameters()]
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        training.train_step(s5, o5, *batch)
    torch.cuda.synchronize()
    assert all(torch.equal(a, q.detach()) for a, q in zip(before, s5.parameters())), "the tainted update reached the weights"
    assert all(float(st["step"]) == 0 for st in o5.state.values())
    with pytest.warns(RuntimeWarning, match="update was skipped"):
        l5 = training.train_step(s5, o5, *batch)
    torch.cuda.synchronize()
    assert float(l5) == float(l2) and MN.precision_of(s5.nerf_fine) == "fp32"
    changed = sum(not torch.equal(a, q.detach()) for a, q in zip(before, s5.parameters()))
    assert changed >= len(before) // 2 and all(float(st["step"]) == 1 for st in o5.state.values())
    assert all(bool(torch.isfinite(q).all()) for q in s5.parameters())
    # a clean model in the default mode: every step updates
    s6 = M.NeRFSystem(training.default_hparams()).to(DEV)
    o6 = torch.optim.Adam(list(s6.parameters()), lr=5e-4, fused=True)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        for _ in range(2):
            training.train_step(s6, o6, *batch)
    assert all(float(st["step"]) == 2 for st in o6.state.values())
    # NeRFSystem.forward on its own (validation, custom loops) checks synchronously and repeats the forward
    s3 = system()
    for k, v in hp.items():
        setattr(s3.hparams, k, v)
    with pytest.warns(RuntimeWarning, match="affected work is repeated"):
        with torch.no_grad():
            r3 = s3(batch[0], {"mirror_mask": batch[2], "is_eval": False, "train_geometry_stage": False})
    with torch.no_grad():
        r2 = s2(batch[0], {"mirror_mask": batch[2], "is_eval": False, "train_geometry_stage": False})
    assert torch.equal(r3["rgb_fine"], r2["rgb_fine"])


def test_guard_costs_no_accuracy_on_clean_models():
    """verify_split (both arithmetics on the model's own samples) still reports fp32-noise-class differences."""
    from mirror_nerf_amd import mirror_nerf as MN
    from mirror_nerf_amd import synthetic as SY
    d = MN.verify_split(_models(SY.OPAQUE)["fine"])
    assert max(d.values()) < 2e-5, d
    assert np.isfinite(list(d.values())).all()


def test_one_backward_overflow_costs_one_adaptation_on_the_static_route():
    """on the static route (one rank, skip mode) the guard flags of step N are read after step N+1 has been queued --
    with the OLD gradient scale, so step N+1 overflows as well.  Its token remembers the scale it was issued with and says nothing
    about the new one: ONE overflow event lowers the scale ONCE (2^-4), not twice.  Trained weights (fixture G11), TotalLoss, batch
    284 of the analytic scene (tests/golden/make_golden_spike.py), whose backward outgrows the f16 range at the default scale."""
    import sys
    from types import SimpleNamespace
    import mirror_nerf_amd as M
    from mirror_nerf_amd import mirror_nerf as MN
    from mirror_nerf_amd import training as T
    from tests.golden import fixtures as FX
    sys.path.insert(0, FX.HERE)
    import make_golden_trained as SC
    if not MN.PRECISION.startswith("split"):
        pytest.skip("the gradient scale belongs to the split arithmetic")
    system = M.NeRFSystem(T.default_hparams(N_importance=64, perturb=0.0, noise_std=0.0))
    z = np.load(f"{FX.HERE}/g11_trained_weights.npz")
    for name, mod in (("coarse", system.nerf_coarse), ("fine", system.nerf_fine)):
        mod.load_state_dict({k[len(name) + 2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(name + "__")})
    system.to(DEV)
    rays, rgbs, masks = (torch.from_numpy(a).to(DEV) for a in SC.scene_views(48, 100, 100))
    opt = T.FlatAdam(list(system.models.values()), lr=1e-6)
    loss_fn = T.total_loss_fn(SimpleNamespace(model_type="nerf"), epoch=5)
    idx = torch.from_numpy(np.random.RandomState(284).randint(rays.shape[0], size=1024)).to(DEV)
    batch = (rays[idx].contiguous(), rgbs[idx].contiguous(), masks[idx].contiguous())
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        for _ in range(5):      # the same overflowing batch five times: trip, (stale trip), then clean at the lowered scale
            T.train_step(system, opt, *batch, loss_fn, epoch=5, gt_valid=True)
        torch.cuda.synchronize()
        T.train_step(system, opt, *batch, loss_fn, epoch=5, gt_valid=True)
    reds = [m.__dict__.get("_mnrf_seed_reduction", 0) for m in system.models.values()]
    assert max(reds) == 4, reds
    assert all(MN.precision_of(m) == "split" for m in system.models.values())
    assert sum("gradient scale is lowered" in str(w.message) for w in caught) == sum(r == 4 for r in reds)
    assert 1 <= sum(int(t.item()) for t in opt._skipped) <= 4      # (the tripping step and the one queued behind it, per model)
