"""GPU tests of the STATIC training route (round 5): the reflected-ray count stays on the device, every nested launch is sized
for the batch and takes the count as its live row count (`_n` entry points of include/mnrf.h), and the whole step can be
captured as one hipGraph (training.GraphedTrainStep).  Yardstick: the host-driven route of the same package -- the one the
fixtures G6 / G9 / G9b / G11 / G16 pin against the reference (train.py:129-348) -- on identical weights, rays and draws."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _system(seed=0, **hp):
    import mirror_nerf_amd as M
    from mirror_nerf_amd import training as T
    torch.manual_seed(seed)
    system = M.NeRFSystem(T.default_hparams(**hp)).to(DEV)
    with torch.no_grad():       # opaque density, a mirror head that straddles 0.5 at the nested level
        for m in system.models.values():
            m.sigma.weight.mul_(20.0)
            m.sigma.bias.fill_(1.0)
            m.is_mirror_net[2].weight.mul_(40.0)
    return system


def _batch(n=256, frac=0.25, seed=3):
    from oracle import mirror_nerf_oracle as O
    g = torch.Generator(device=DEV)
    g.manual_seed(seed)
    rays = torch.from_numpy(O.synthetic_rays(64, 64)[:: max(1, 4096 // n)][:n].copy()).to(DEV)
    target = torch.rand(n, 3, device=DEV, generator=g)
    gt = (torch.rand(n, device=DEV, generator=g) < frac).float()
    return rays, target, gt


def _grads(system):
    from mirror_nerf_amd.weights import params_of
    return [None if q.grad is None else q.grad.detach().clone() for m in system.models.values() for q in params_of(m)]


def _fwd_bwd(system, rays, target, gt, static, loss="color_mask", gt_valid=True):
    from mirror_nerf_amd import training as T
    ex = dict(T.extra_info(system.hparams, gt, 5), _guard=False)
    if static:
        ex.update(_static=True, _gt_valid=gt_valid)
    system.zero_grad(set_to_none=True)
    res = system(rays, ex)
    fn = T.total_loss_fn() if loss == "total" else T.color_mask_loss
    val = fn(res, target, gt, rays) if getattr(fn, "needs_rays", False) else fn(res, target, gt)
    val.backward()
    return res, float(val), _grads(system)


def _cmp_grads(a, b, tol):
    worst = 0.0
    for x, y in zip(a, b):
        assert (x is None) == (y is None)
        if x is None:
            continue
        scale = float(y.abs().max()) + 1e-30
        worst = max(worst, float((x - y).abs().max()) / scale)
    assert worst <= tol, worst
    return worst


@pytest.mark.parametrize("frac", [0.25, 0.0, 1.0])
@pytest.mark.parametrize("loss", ["color_mask", "total"])
def test_static_route_equals_host_route(frac, loss):
    """Values of every per-ray map bit for bit (the same kernels see the same rows), gradients to the summation order of the
    weight-gradient GEMM (its work plan deals the 32-sample stages to workgroups differently when it is made on the device for a
    launch of one workgroup per CU)."""
    system = _system(perturb=0.0, noise_std=0.0)
    rays, target, gt = _batch(frac=frac)
    res_h, loss_h, g_h = _fwd_bwd(system, rays, target, gt, static=False, loss=loss)
    res_s, loss_s, g_s = _fwd_bwd(system, rays, target, gt, static=True, loss=loss)
    for k in ("rgb_coarse", "rgb_fine", "depth_fine", "mirror_mask_fine", "opacity_fine", "x_surface_fine", "surface_normal_fine"):
        assert torch.equal(res_h[k], res_s[k]), k
    if loss_h != loss_h:      # TotalLoss over a batch without a mirror ray is NaN on both routes (a mean over an empty selection,
        assert loss_s != loss_s     # losses.py:150-173 does the same): nothing to compare beyond that
        return
    assert loss_h == loss_s
    _cmp_grads(g_s, g_h, 2e-5)


def test_static_route_with_two_bounces_and_predicted_masks():
    """max_recursive_level = 2: the second level compacts by the PREDICTED mask, thresholded in place with a live row count, and
    hands its own count to the third render; invalid ground truth at level 0 takes the predicted mask there too (train.py:153-166)."""
    system = _system(perturb=0.0, noise_std=0.0, max_recursive_level=2)
    rays, target, gt = _batch(n=192, frac=0.5)
    for gt_valid, mask in ((True, gt), (False, torch.full_like(gt, -1.0))):
        res_h, loss_h, g_h = _fwd_bwd(system, rays, target, mask, static=False)
        res_s, loss_s, g_s = _fwd_bwd(system, rays, target, mask, static=True, gt_valid=gt_valid)
        for k in ("rgb_coarse", "rgb_fine", "mirror_mask_fine"):
            assert torch.equal(res_h[k], res_s[k]), (k, gt_valid)
        assert loss_h == loss_s
        _cmp_grads(g_s, g_h, 2e-5)


def test_static_route_without_compaction():
    """only_trace_rays_in_mirrors = False: every ray is reflected as soon as one is a mirror (train.py:248-259); the static route
    reflects them always -- with no mirror pixel the blend leaves every colour as it is."""
    system = _system(perturb=0.0, noise_std=0.0, only_trace_rays_in_mirrors=False)
    for frac in (0.3, 0.0):
        rays, target, gt = _batch(n=128, frac=frac)
        res_h, loss_h, g_h = _fwd_bwd(system, rays, target, gt, static=False)
        res_s, loss_s, g_s = _fwd_bwd(system, rays, target, gt, static=True)
        assert torch.equal(res_h["rgb_fine"], res_s["rgb_fine"])
        assert loss_h == loss_s
        _cmp_grads(g_s, g_h, 2e-5)


def test_graphed_step_takes_zero_and_all_mirror_batches_through_one_graph():
    """One capture, three replays: a batch with NO mirror ray, one with ALL rays mirrors, a mixed one.  After every replay the
    weights equal those of a twin system stepped by the host-driven train_step on the same batches (same draws: perturb and
    noise off) to the accuracy of one Adam step's arithmetic."""
    from mirror_nerf_amd import training as T
    from mirror_nerf_amd.weights import params_of
    a = _system(perturb=0.0, noise_std=0.0)
    b = _system(perturb=0.0, noise_std=0.0)       # (same seed: the same weights)
    opt_a = T.FlatAdam(list(a.models.values()), lr=5e-4)
    opt_b = T.FlatAdam(list(b.models.values()), lr=5e-4)
    n = 256
    step = T.GraphedTrainStep(a, opt_a, n, gt_valid=True)
    graphs = []
    for i, frac in enumerate((0.0, 1.0, 0.3, 0.0)):
        rays, target, gt = _batch(n=n, frac=frac, seed=10 + i)
        la = step(rays, target, gt)
        graphs.append(step.graph)
        lb = T.train_step(b, opt_b, rays, target, gt)
        torch.cuda.synchronize()
        assert abs(float(la) - float(lb)) <= 1e-6 * max(1.0, abs(float(lb))), (i, float(la), float(lb))
        for qa, qb in zip((q for m in a.models.values() for q in params_of(m)), (q for m in b.models.values() for q in params_of(m))):
            # Adam's first steps move every weight by ~lr whatever the gradient's size: a gradient entry that differs in its last
            # bits near zero can flip a whole lr step, so compare against lr
            assert float((qa - qb).abs().max()) <= 2.5e-4 * (i + 1), i
    assert all(g is graphs[0] for g in graphs) and not step.ended
    assert opt_a._calls == 4 and int(opt_a._step_dev.item()) == 4


def test_graphed_step_follows_the_learning_rate_and_counts_steps():
    from mirror_nerf_amd import training as T
    a = _system(perturb=1.0, noise_std=1.0)
    opt = T.FlatAdam(list(a.models.values()), lr=5e-4)
    step = T.GraphedTrainStep(a, opt, 128, gt_valid=True)
    rays, target, gt = _batch(n=128)
    w0 = opt.flats[0].detach().clone()
    step(rays, target, gt)
    torch.cuda.synchronize()
    d1 = float((opt.flats[0] - w0).abs().max())
    assert 1e-4 < d1 <= 5.1e-4          # Adam's first step: |update| <= lr
    opt.param_groups[0]["lr"] = 0.0      # a scheduler's move
    w1 = opt.flats[0].detach().clone()
    step(rays, target, gt)
    torch.cuda.synchronize()
    assert float((opt.flats[0] - w1).abs().max()) == 0.0
    assert int(opt._step_dev.item()) == 2 and int(opt._skipped[0].item()) == 0


def test_graphed_step_guard_trip_skips_the_update_and_ends_the_graph():
    """A weight beyond the f16 range trips the guard inside the captured step: the update is skipped on the device, the host
    learns it at the next call, pins the models and continues on train_step."""
    from mirror_nerf_amd import training as T
    from mirror_nerf_amd.mirror_nerf import precision_of
    a = _system(perturb=0.0, noise_std=0.0)
    opt = T.FlatAdam(list(a.models.values()), lr=5e-4)
    step = T.GraphedTrainStep(a, opt, 128, gt_valid=True)
    rays, target, gt = _batch(n=128)
    step(rays, target, gt)
    torch.cuda.synchronize()
    with torch.no_grad():
        a.nerf_fine.xyz_encoding_3[0].weight[0, 0] = 1e6        # (through the flat tensor's view; the replay re-packs it)
    w = opt.flats[1].detach().clone()
    step(rays, target, gt)
    torch.cuda.synchronize()
    assert torch.equal(opt.flats[1], w) and int(opt._skipped[1].item()) == 1
    with pytest.warns(RuntimeWarning):
        step(rays, target, gt)
    assert step.ended and precision_of(a.nerf_fine) == "fp32"


# ----------------------------------------------------------------------------- the `_n` entry points themselves
def test_live_row_count_leaves_the_rows_past_it_alone():
    """mnrf_sample_coarse_n / mnrf_composite_n / mnrf_sample_fine_n with n_live < capacity: the live rows equal the namesake's, the
    rest of every output keeps its sentinel."""
    from mirror_nerf_amd import _lib
    L, p = _lib.lib(), _lib.ptr
    torch.manual_seed(0)
    N, S, live = 70, 64, 23
    n_live = torch.tensor([live], dtype=torch.int32, device=DEV)
    rays = torch.randn(N, 8, device=DEV)
    rays[:, 6], rays[:, 7] = 0.1, 6.0
    zs = torch.linspace(0, 1, S).to(DEV)
    z_a, z_b = torch.full((N, S), -7.0, device=DEV), torch.full((N, S), -7.0, device=DEV)
    _lib.check(L.mnrf_sample_coarse(p(rays), N, p(zs), S, 0, 0.0, None, p(z_a), _lib.stream()), "a")
    _lib.check(L.mnrf_sample_coarse_n(p(rays), N, p(zs), S, 0, 0.0, None, p(z_b), p(n_live), _lib.stream()), "b")
    assert torch.equal(z_a[:live], z_b[:live]) and bool((z_b[live:] == -7.0).all())
    sigma, rgb = torch.randn(N, S, device=DEV) * 3, torch.rand(N * S, 3, device=DEV)
    outs = {}
    for tag, nl in (("a", None), ("b", n_live)):
        w, op, rm, d, xs = (torch.full(s, -7.0, device=DEV) for s in ((N, S), (N,), (N, 3), (N,), (N, 3)))
        _lib.check(L.mnrf_composite_n(p(rays), N, S, p(sigma), p(z_a), None, p(rgb), None, None, None, 0, p(w), p(op), p(rm), p(d),
                                      None, None, None, None, p(xs), p(nl), _lib.stream()), "composite")
        outs[tag] = (w, op, rm, d, xs)
    for ta, tb in zip(outs["a"], outs["b"]):
        assert torch.equal(ta[:live], tb[:live]) and bool((tb[live:] == -7.0).all())
    u = torch.linspace(0, 1, 128).to(DEV)
    f_a, f_b = torch.full((N, S + 128), -7.0, device=DEV), torch.full((N, S + 128), -7.0, device=DEV)
    _lib.check(L.mnrf_sample_fine_n(p(z_a), p(outs["a"][0]), N, S, p(u), 0, 128, p(f_a), None, _lib.stream()), "fine a")
    _lib.check(L.mnrf_sample_fine_n(p(z_a), p(outs["a"][0]), N, S, p(u), 0, 128, p(f_b), p(n_live), _lib.stream()), "fine b")
    assert torch.equal(f_a[:live], f_b[:live]) and bool((f_b[live:] == -7.0).all())
    zero = torch.zeros(1, dtype=torch.int32, device=DEV)       # nothing live: nothing written
    f_c = torch.full((N, S + 128), -7.0, device=DEV)
    _lib.check(L.mnrf_sample_fine_n(p(z_a), p(outs["a"][0]), N, S, p(u), 0, 128, p(f_c), p(zero), _lib.stream()), "fine c")
    assert bool((f_c == -7.0).all())


def test_device_made_gemm_plan_gives_the_host_plan_gradients():
    """mnrf_dw_planes2_n (work plan made on the device from live counts) against mnrf_dw_planes2 on the same planes: a field
    evaluation of 300 rays x 64 samples of which 117 rays are live, plus one whose count is zero."""
    from mirror_nerf_amd import _lib
    from mirror_nerf_amd.autograd import FieldFn, _Pending
    from mirror_nerf_amd.weights import params_of
    system = _system(perturb=0.0, noise_std=0.0)
    model = system.nerf_coarse
    torch.manual_seed(1)
    N, S = 300, 64
    rays = torch.randn(N, 8, device=DEV) * 0.5
    z = torch.sort(torch.rand(N, S, device=DEV) * 4 + 0.1, 1)[0]
    de = torch.randn(N, 27, device=DEV)

    def run(n_live, rows):
        system.zero_grad(set_to_none=True)
        want = (False, 0, None, n_live) if n_live is not None else False
        sigma, rgb, pn, mir, _ = FieldFn.apply(model, S, None, rays[:rows].contiguous(), z[:rows].contiguous(), de[:rows].contiguous(),
                                               want, *params_of(model))
        live = rows if n_live is None else int(n_live.item())
        (sigma.view(rows, S)[:live].sum() + (rgb.view(rows, S, 3)[:live] ** 2).sum() + pn.view(rows, S, 3)[:live, :, 0].sum()
         + mir.view(rows, S)[:live].sum()).backward()
        return [q.grad.detach().clone() for q in params_of(model)]
    ref = run(None, 117)
    got = run(torch.tensor([117], dtype=torch.int32, device=DEV), N)
    _cmp_grads(got, ref, 2e-5)
    none = run(torch.tensor([0], dtype=torch.int32, device=DEV), N)
    assert all(float(g.abs().max()) == 0.0 for g in none)


def test_flat_adam_refuses_orphaned_parameters_and_skips_modules_without_gradients():
    """FlatAdam re-points every parameter at a view of its flat tensor.  A parameter whose storage is replaced later
    (`.to()`, `.float()`, load_state_dict(assign=True)) would leave step() updating an orphaned tensor: step() raises instead.  A
    module no backward pass reached is skipped like torch.optim.Adam skips parameters without .grad (no momentum-only update)."""
    from mirror_nerf_amd import training as T
    a = _system(perturb=0.0, noise_std=0.0)
    opt = T.FlatAdam(list(a.models.values()), lr=5e-4)
    rays, target, gt = _batch(n=64)
    T.train_step(a, opt, rays, target, gt)                      # builds momentum
    torch.cuda.synchronize()
    w = [fp.detach().clone() for fp in opt.flats]
    m = [t.clone() for t in opt._m]
    opt.zero_grad()
    opt.step()                                                   # no gradients anywhere: nothing moves, the moments do not decay
    torch.cuda.synchronize()
    assert all(torch.equal(x, fp.detach()) for x, fp in zip(w, opt.flats)) and all(torch.equal(x, y) for x, y in zip(m, opt._m))
    q = a.nerf_fine.is_mirror_net[2].bias
    q.data = q.data.clone()                                      # what a cast / move / re-assignment does to the aliasing
    with pytest.raises(RuntimeError, match="no longer aliases"):
        opt.step()


# ---------------------------------------------------------------- launches merged in the second half of round 5
def test_ray_prologue_equals_embed_and_sample_coarse():
    """mnrf_ray_prologue_n = mnrf_embed_n of the direction columns + mnrf_sample_coarse_n, bit for bit, with and without a live
    row count (rows past it are not written)."""
    from mirror_nerf_amd import _lib
    L, p = _lib.lib(), _lib.ptr
    g = torch.Generator(device=DEV)
    g.manual_seed(11)
    n, ns, nf = 200, 64, 4
    rays = torch.randn(n, 8, device=DEV, generator=g)
    rays[:, 6], rays[:, 7] = 0.3, 5.0
    steps = torch.linspace(0, 1, ns, device=DEV)
    prand = torch.rand(n, ns, device=DEV, generator=g)
    for use_disp, perturb in ((0, 0.0), (0, 1.0), (1, 1.0)):
        for live in (None, 77):
            nl = None if live is None else torch.tensor([live], dtype=torch.int32, device=DEV)
            want_e, want_z = torch.full((n, 27), -7.0, device=DEV), torch.full((n, ns), -7.0, device=DEV)
            got_e, got_z = want_e.clone(), want_z.clone()
            dirs = rays[:, 3:6].contiguous()
            _lib.check(L.mnrf_embed_n(p(dirs), n, 3, nf, p(want_e), p(nl), _lib.stream()), "embed")
            _lib.check(L.mnrf_sample_coarse_n(p(rays), n, p(steps), ns, use_disp, perturb, p(prand), p(want_z), p(nl), _lib.stream()), "coarse")
            _lib.check(L.mnrf_ray_prologue_n(p(rays), n, nf, p(steps), ns, use_disp, perturb, p(prand), p(got_e), p(got_z), p(nl),
                                             _lib.stream()), "prologue")
            assert torch.equal(got_e, want_e) and torch.equal(got_z, want_z)
    assert L.mnrf_ray_prologue_n(p(rays), n, nf, p(steps), 2, 0, 0.0, None, p(got_e), p(got_z), None, _lib.stream()) < 0     # n_samples >= 3


@pytest.mark.parametrize("absent", [(), (1,), (0, 2, 3), (4,), (0, 1, 2, 3)])
def test_ray_fan_backward_is_autograds_sum(absent):
    """mnrf_ray_fan_backward_n against the torch ops it replaces, in autograd's order: pairwise adds of the four ray gradients, the
    encoding's backward on the sum of its two gradients, padded to eight columns, added last.  Bit for bit; any piece may be absent."""
    from mirror_nerf_amd import _lib
    L, p = _lib.lib(), _lib.ptr
    g = torch.Generator(device=DEV)
    g.manual_seed(5)
    n, nf = 300, 4
    rays = torch.randn(n, 8, device=DEV, generator=g)
    gs = [torch.randn(n, 8, device=DEV, generator=g) for _ in range(4)]
    ga, gb = torch.randn(n, 27, device=DEV, generator=g), torch.randn(n, 27, device=DEV, generator=g)
    pieces = [None if k in absent else t for k, t in enumerate(gs)]
    da = None if 4 in absent else ga
    db = None if 5 in absent else gb
    want = None
    for t in pieces:
        if t is not None:
            want = t.clone() if want is None else want + t
    if da is not None or db is not None:
        gsum = da if db is None else (db if da is None else da + db)
        gx = torch.empty(n, 3, device=DEV)
        _lib.check(L.mnrf_embed_backward_n(p(rays[:, 3:6].contiguous()), p(gsum.contiguous()), n, 3, nf, p(gx), None, _lib.stream()), "embed bwd")
        pad = torch.zeros(n, 8, device=DEV)
        pad[:, 3:6] = gx
        want = pad if want is None else want + pad
    got = torch.empty(n, 8, device=DEV)
    _lib.check(L.mnrf_ray_fan_backward_n(p(pieces[0]), p(pieces[1]), p(pieces[2]), p(pieces[3]), p(rays), p(da), p(db), n, nf, p(got), None,
                                         _lib.stream()), "fan")
    assert torch.equal(got, want)
    nl = torch.tensor([123], dtype=torch.int32, device=DEV)
    got2 = torch.full((n, 8), -3.0, device=DEV)
    _lib.check(L.mnrf_ray_fan_backward_n(p(pieces[0]), p(pieces[1]), p(pieces[2]), p(pieces[3]), p(rays), p(da), p(db), n, nf, p(got2), p(nl),
                                         _lib.stream()), "fan (live rows)")
    assert torch.equal(got2[:123], want[:123]) and bool((got2[123:] == -3.0).all())


def test_ray_fan_backward_refuses_nothing():
    from mirror_nerf_amd import _lib
    out = torch.empty(4, 8, device=DEV)
    assert _lib.lib().mnrf_ray_fan_backward_n(None, None, None, None, None, None, None, 4, 4, _lib.ptr(out), None, _lib.stream()) < 0


def test_pack_weights_n_equals_one_by_one():
    """mnrf_pack_weights_n (both models of a step in one launch pair) writes the images mnrf_pack_weights writes."""
    import mirror_nerf_amd as M
    from mirror_nerf_amd.weights import pack_state, pack_states, param_refs
    torch.manual_seed(3)
    models = [M.MirrorNeRF().to(DEV) for _ in range(5)]          # five: more than one batch of four
    states = [{full: sub._parameters[pname] for sub, pname, full in param_refs(m)} for m in models]
    from mirror_nerf_amd import _lib
    blank = lambda: torch.zeros(_lib.lib().mnrf_packed_floats(), device=DEV)  # noqa: E731  (regions no tuning writes stay as they were)
    one = [pack_state(s, blank()) for s in states]
    many = pack_states(states, [blank() for _ in states])
    for a, b in zip(one, many):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    with torch.no_grad():
        models[1].sigma.weight[0, 0] = float("inf")             # the weight bit of the range guard belongs to ITS image only
    many = pack_states(states, many)
    words = [int(t[-1:].view(torch.int32).item()) for t in many]
    assert words[1] != 0 and all(w == 0 for k, w in enumerate(words) if k != 1)


def test_composite_sample_equals_composite_then_sample_fine():
    """mnrf_composite_sample_n = mnrf_composite_n followed by mnrf_sample_fine_n on its weights, bit for bit (maps, weights, depths),
    with a ragged last workgroup and with a live row count."""
    from mirror_nerf_amd import _lib
    L, p = _lib.lib(), _lib.ptr
    g = torch.Generator(device=DEV)
    g.manual_seed(21)
    n, S, n_imp = 203, 64, 64
    f = lambda *s: torch.empty(*s, device=DEV)  # noqa: E731
    rays = torch.randn(n, 8, device=DEV, generator=g)
    z = torch.sort(torch.rand(n, S, device=DEV, generator=g) * 4 + 0.5, dim=1)[0].contiguous()
    sigma = torch.randn(n, S, device=DEV, generator=g) * 3
    noise = torch.randn(n, S, device=DEV, generator=g)
    rgb, pn, gn = (torch.rand(n * S, 3, device=DEV, generator=g) for _ in range(3))
    mir = torch.rand(n * S, device=DEV, generator=g)
    for u in (torch.rand(n, n_imp, device=DEV, generator=g), torch.linspace(0, 1, n_imp, device=DEV)):
        for live in (None, 150):
            nl = None if live is None else torch.tensor([live], dtype=torch.int32, device=DEV)
            outs = []
            for fused in (False, True):
                o = dict(w=f(n, S).fill_(-5), op=f(n).fill_(-5), c=f(n, 3).fill_(-5), d=f(n).fill_(-5), m=f(n).fill_(-5), sn=f(n, 3).fill_(-5),
                         sg=f(n, 3).fill_(-5), nd=f(n).fill_(-5), xs=f(n, 3).fill_(-5), zf=f(n, S + n_imp).fill_(-5))
                common = (p(rays), n, S, p(sigma), p(z), p(noise), p(rgb), p(mir), p(pn), p(gn), 0, p(o["w"]), p(o["op"]), p(o["c"]), p(o["d"]),
                          p(o["m"]), p(o["sn"]), p(o["sg"]), p(o["nd"]), p(o["xs"]))
                if fused:
                    _lib.check(L.mnrf_composite_sample_n(*common, p(u), 1 if u.dim() == 2 else 0, n_imp, p(o["zf"]), p(nl), _lib.stream()), "fused")
                else:
                    _lib.check(L.mnrf_composite_n(*common, p(nl), _lib.stream()), "composite")
                    _lib.check(L.mnrf_sample_fine_n(p(z), p(o["w"]), n, S, p(u), 1 if u.dim() == 2 else 0, n_imp, p(o["zf"]), p(nl), _lib.stream()),
                               "sample_fine")
                outs.append(o)
            for k in outs[0]:
                assert torch.equal(outs[0][k], outs[1][k]), (k, live)
            rows = n if live is None else live
            assert bool((outs[1]["zf"][:rows] > 0).all()) and (live is None or bool((outs[1]["zf"][rows:] == -5).all()))


# ------------------------------------------------------------------------------------------ round 6: run.sh:259-280's recipe
def _recipe_loss(stage, epoch):
    from types import SimpleNamespace
    from mirror_nerf_amd import training as T
    return T.total_loss_fn(SimpleNamespace(use_plane_consistent_loss=True), epoch=epoch, train_geometry_stage=stage)


def _fwd_bwd_recipe(system, rays, target, gt, static, u, stage=False, epoch=5, gt_valid=True):
    """One forward + TotalLoss (all five terms, --use_plane_consistent_loss) + backward with the plane term's uniform numbers
    injected, so that the host route (one read of the mirror-ray count, picks formed on the host) and the static route (count and
    picks on the device) draw the same quadruples."""
    from mirror_nerf_amd import training as T
    from mirror_nerf_amd.losses import get_loss
    from types import SimpleNamespace
    system.train_geometry_stage = stage
    ex = dict(T.extra_info(system.hparams, gt, epoch, stage), _guard=False)
    if static:
        ex.update(_static=True, _gt_valid=gt_valid)
    system.zero_grad(set_to_none=True)
    tgt = T.stage_target(system.hparams, target, gt, gt_valid) if stage else target
    res = system(rays, ex)
    crit = get_loss(SimpleNamespace(use_plane_consistent_loss=True))
    batch = {"rgbs": tgt, "mirror_mask": gt, "rays": rays, "_plane_u": u}
    if static:
        batch["_plane_on_device"] = True
    total, terms = crit(res, batch, train_geometry_stage=stage, epoch=epoch)
    total.backward()
    return {k: float(v) for k, v in terms.items()}, float(total), _grads(system)


@pytest.mark.parametrize("stage,epoch", [(False, 5), (True, 2)])
def test_plane_consistent_loss_same_draws_on_host_and_static_route(stage, epoch):
    """losses.py:80-127 without a host read.  Same uniform numbers on both routes -> the same quadruples -> every
    term and the total bit for bit; gradients to the summation order of the GEMM (and of the plane term's atomics)."""
    system = _system(perturb=0.0, noise_std=0.0)
    rays, target, gt = _batch(n=256, frac=0.3)
    g = torch.Generator(device=DEV)
    g.manual_seed(77)
    u = torch.rand(2, 4 * (256 // 4), device=DEV, generator=g)
    terms_h, loss_h, g_h = _fwd_bwd_recipe(system, rays, target, gt, False, u, stage, epoch)
    terms_s, loss_s, g_s = _fwd_bwd_recipe(system, rays, target, gt, True, u, stage, epoch)
    assert set(terms_h) == set(terms_s) == {"color_loss", "mirror_mask_loss", "plane_consistent_loss", "normal_loss", "normal_reg_loss"}
    assert terms_h["plane_consistent_loss"] > 0.0
    for k in terms_h:
        assert terms_h[k] == terms_s[k], (k, terms_h[k], terms_s[k])
    assert loss_h == loss_s
    _cmp_grads(g_s, g_h, 2e-5)
    system.train_geometry_stage = False


def test_plane_term_on_the_device_switches_off_for_invalid_gt_and_for_fewer_than_four_mirror_rays():
    """losses.py:116-119 (an invalid GT entry: no mask, no term) and losses.py:98-100 (times = M // 4 = 0) decided by the kernel from the
    count words: the term is exactly 0 and x_surface receives no gradient from it."""
    import mirror_nerf_amd as M
    from types import SimpleNamespace
    n = 64
    g = torch.Generator(device=DEV)
    g.manual_seed(5)
    res = {f"rgb_{t}": torch.rand(n, 3, device=DEV, generator=g) for t in ("coarse", "fine")}
    xs = {t: torch.randn(n, 3, device=DEV, generator=g).requires_grad_(True) for t in ("coarse", "fine")}
    res.update({f"x_surface_{t}": xs[t] * 1.0 for t in xs})
    crit = M.get_loss(SimpleNamespace(use_plane_consistent_loss=True))
    rays = torch.zeros(n, 8, device=DEV)
    for name, gt in (("invalid", torch.cat([torch.full((1,), -1.0, device=DEV), torch.ones(n - 1, device=DEV)])),
                     ("three mirror rays", torch.cat([torch.ones(3, device=DEV), torch.zeros(n - 3, device=DEV)]))):
        total, terms = crit(res, {"rgbs": torch.rand(n, 3, device=DEV, generator=g), "mirror_mask": gt, "rays": rays,
                                  "_plane_on_device": True}, train_geometry_stage=False, epoch=5)
        assert float(terms["plane_consistent_loss"]) == 0.0, name
    gt = (torch.rand(n, device=DEV, generator=g) < 0.5).float()
    total, terms = crit(res, {"rgbs": torch.rand(n, 3, device=DEV, generator=g), "mirror_mask": gt, "rays": rays, "_plane_on_device": True},
                        train_geometry_stage=False, epoch=5)
    assert float(terms["plane_consistent_loss"]) > 0.0
    total.backward()
    # only mirror rows receive a gradient from the plane term (the colour term does not reach x_surface)
    for t in xs:
        assert float(xs[t].grad[gt == 0].abs().max()) == 0.0 and float(xs[t].grad[gt != 0].abs().max()) > 0.0


def test_plane_term_on_the_device_against_the_oracle():
    """The picks the kernel forms from its uniform numbers, restated on the host (floor(u * M) in fp32, clamped), handed to the
    oracle's PlaneConsistentLoss (losses.py:88-110 restated): value to 2e-6, gradient of x_surface to 1e-6 of its largest entry --
    the bar tests/test_loss.py holds the host route to against the reference's captured draws (G10)."""
    import numpy as np
    import mirror_nerf_amd as M
    from oracle import mirror_nerf_oracle as O
    from types import SimpleNamespace
    n = 200
    rs = np.random.RandomState(4)
    res_np = {f"rgb_{t}": rs.uniform(size=(n, 3)).astype(np.float32) for t in ("coarse", "fine")}
    res_np.update({f"x_surface_{t}": rs.normal(size=(n, 3)).astype(np.float32) for t in ("coarse", "fine")})
    gt = (rs.uniform(size=n) < 0.4).astype(np.float32)
    m = int(gt.sum())
    u = rs.uniform(size=(2, 4 * (n // 4))).astype(np.float32)
    batch_np = {"rgbs": rs.uniform(size=(n, 3)).astype(np.float32), "mirror_mask": gt, "rays": np.zeros((n, 8), np.float32)}
    picks = {t: np.minimum((u[k, :4 * (m // 4)] * np.float32(m)).astype(np.int64), m - 1).reshape(m // 4, 4) for k, t in enumerate(("fine", "coarse"))}
    hp = dict(O.LOSS_DEFAULTS, use_plane_consistent_loss=True)
    want = float(O.plane_consistent_loss({k: v.copy() for k, v in res_np.items()}, batch_np, hp, picks))
    t_in = {k: torch.from_numpy(v.copy()).to(DEV).requires_grad_(k.startswith("x_surface")) for k, v in res_np.items()}
    crit = M.get_loss(SimpleNamespace(use_plane_consistent_loss=True, color_loss_weight=0.0))
    tb = {k: torch.from_numpy(v).to(DEV) for k, v in batch_np.items()}
    tb.update(_plane_on_device=True, _plane_u=torch.from_numpy(u).to(DEV))
    total, terms = crit({k: v * 1.0 for k, v in t_in.items()}, tb, train_geometry_stage=False, epoch=5)
    got = float(terms["plane_consistent_loss"])
    assert abs(got - want) <= 2e-6 * max(1.0, abs(want)), (got, want)
    total.backward()
    # finite-difference check of the stored gradient along a random direction (the oracle has no gradients)
    d = {t: rs.normal(size=(n, 3)).astype(np.float32) for t in ("coarse", "fine")}
    eps = 1e-3
    plus = {k: v + eps * d[k[10:]] if k.startswith("x_surface") else v for k, v in res_np.items()}
    minus = {k: v - eps * d[k[10:]] if k.startswith("x_surface") else v for k, v in res_np.items()}
    fd = (float(O.plane_consistent_loss(plus, batch_np, hp, picks)) - float(O.plane_consistent_loss(minus, batch_np, hp, picks))) / (2 * eps)
    an = sum(float((t_in[f"x_surface_{t}"].grad.cpu().numpy().astype(np.float64) * d[t]).sum()) for t in d)
    assert abs(fd - an) <= 2e-3 * max(abs(fd), 1e-3), (fd, an)


def test_static_route_refuses_the_level_zero_mask_on_compacted_rays():
    """--detach_density_outside_mirror_for_mask_loss with compacted reflected rays.  The reference (and the host
    route) fail with an IndexError at the nested level (train.py:253-259 hands the un-compacted mask on); on the static route the
    compacted rays keep the chunk's capacity, so the shapes agree -- it must refuse all the same instead of steering the wrong rows."""
    system = _system(perturb=0.0, noise_std=0.0, detach_density_outside_mirror_for_mask_loss=True)
    rays, target, gt = _batch(n=128, frac=0.3)
    with pytest.raises(IndexError):
        _fwd_bwd(system, rays, target, gt, static=False)
    with pytest.raises(IndexError):
        _fwd_bwd(system, rays, target, gt, static=True)
    # without compaction the rows of the nested level ARE the rows of the mask: both routes run and agree
    system = _system(perturb=0.0, noise_std=0.0, detach_density_outside_mirror_for_mask_loss=True, only_trace_rays_in_mirrors=False)
    _r, loss_h, g_h = _fwd_bwd(system, rays, target, gt, static=False)
    _r, loss_s, g_s = _fwd_bwd(system, rays, target, gt, static=True)
    assert loss_h == loss_s
    _cmp_grads(g_s, g_h, 2e-5)


def _recipe_loss_with_draws(stage, epoch, u_buf):
    """total_loss_fn's shape with the plane term's uniform numbers taken from `u_buf` (a static tensor the test refills per step):
    the captured step forms its picks on the device, the host-driven step on the host -- from the same numbers."""
    from types import SimpleNamespace
    from mirror_nerf_amd.losses import get_loss
    crit = get_loss(SimpleNamespace(use_plane_consistent_loss=True))

    def fn(res, target, gt, rays, static=False):
        batch = {"rgbs": target, "mirror_mask": gt, "rays": rays, "_plane_u": u_buf}
        if static:
            batch["_plane_on_device"] = True
        return crit(res, batch, train_geometry_stage=stage, epoch=epoch)[0]
    fn.needs_rays, fn.takes_static, fn.train_geometry_stage = True, True, stage
    return fn


@pytest.mark.parametrize("stage,epoch", [(False, 5), (True, 2)])
def test_graphed_step_carries_the_run_sh_recipe(stage, epoch):
    """GraphedTrainStep with TotalLoss's five terms (the plane term forming its picks on the device) after and inside the geometry
    stage (run.sh:276-277): captured once, three replays; against train_step on the HOST-DRIVEN route (one read of the mirror-ray
    count, picks formed on the host from the same numbers): the loss bit for bit on the first step, weights after every step to one
    Adam step's arithmetic."""
    from mirror_nerf_amd import training as T
    from mirror_nerf_amd.weights import params_of
    a = _system(perturb=0.0, noise_std=0.0)
    b = _system(perturb=0.0, noise_std=0.0)
    a.train_geometry_stage = b.train_geometry_stage = stage
    opt_a = T.FlatAdam(list(a.models.values()), lr=5e-4)
    opt_b = T.FlatAdam(list(b.models.values()), lr=5e-4)
    n = 256
    u_buf = torch.zeros(2, 4 * (n // 4), device=DEV)
    with pytest.raises(ValueError):       # a loss built for the other stage (train.py:426 and 439-446 hand over the same flag)
        T.GraphedTrainStep(a, opt_a, n, loss_fn=_recipe_loss(not stage, epoch), epoch=epoch)
    step = T.GraphedTrainStep(a, opt_a, n, loss_fn=_recipe_loss_with_draws(stage, epoch, u_buf), epoch=epoch, gt_valid=True)
    fn_b = _recipe_loss_with_draws(stage, epoch, u_buf)
    g = torch.Generator(device=DEV)
    g.manual_seed(123)
    graphs = []
    for i in range(3):
        rays, target, gt = _batch(n=n, frac=0.3, seed=20 + i)
        u_buf.copy_(torch.rand(u_buf.shape, device=DEV, generator=g))
        la = step(rays, target, gt)
        graphs.append(step.graph)
        lb = T.train_step(b, opt_b, rays, target, gt, fn_b, epoch=epoch)
        torch.cuda.synchronize()
        assert not step.ended
        if i == 0:
            assert float(la) == float(lb), (float(la), float(lb))
        assert abs(float(la) - float(lb)) <= 2e-5 * max(1.0, abs(float(lb))), (i, float(la), float(lb))
        for qa, qb in zip((q for m in a.models.values() for q in params_of(m)), (q for m in b.models.values() for q in params_of(m))):
            assert float((qa - qb).abs().max()) <= 2.5e-4 * (i + 1), i
    assert all(gr is graphs[0] for gr in graphs)
    assert opt_a._calls == 3 and int(opt_a._step_dev.item()) == 3 and opt_a._steps == [3, 3]
    # the stage ends (train.py:387-391): other launches -- the next call captures again, and the loss must be the other stage's
    a.train_geometry_stage = b.train_geometry_stage = not stage
    with pytest.raises(ValueError):
        step(rays, target, gt)
    a.train_geometry_stage = b.train_geometry_stage = False


# ------------------------------------------------------------------------------------------ round 6: config 5 on the static route
def test_hash_grid_model_takes_the_static_route():
    """VERDICT r5 "missing" #4 / item 4b: MirrorNeRFTcnn through the recursion's static route (mnrf_tcnn_forward_n / _backward_n: the
    reflected rays keep the batch's capacity, the kernels take the count as their live row count).  Against the host-driven route
    (train.py:102-348 with one device->host read per level) on the same weights and rays: every per-ray map bit for bit, the table
    gradient to 1e-6 of its largest entry (fixed-point scatter on the hashed levels, fp32 atomics on the coarse ones), the MLP
    gradients to the order of their atomics;
    with 25 %, no and only mirror rays; train_step(gt_valid=True) takes the route and updates the weights like the host route."""
    from types import SimpleNamespace
    import mirror_nerf_amd as M
    from mirror_nerf_amd import training as T
    from oracle import mirror_nerf_oracle as O
    hp = SimpleNamespace(model_type="nerf_tcnn", bound=2.0, predict_normal=True, predict_mirror_mask=True, N_samples=24,
                         N_importance=24, use_disp=False, perturb=0, noise_std=0, chunk=4096, only_one_field=False,
                         trace_secondary_rays=True, max_recursive_level=1, only_trace_rays_in_mirrors=True, for_vis=False)

    def system():
        torch.manual_seed(4)
        s = M.NeRFSystem(hp)
        with torch.no_grad():
            for mdl in (s.nerf_coarse, s.nerf_fine):
                mdl.encoder.embeddings.uniform_(-0.05, 0.05)
                mdl.encoder.embeddings[int(mdl.cfg["offsets"][5]):] = 0      # (smooth field: see test_tcnn_train_recursion_matches_torch_field)
                mdl.sigma_net[1].weight[0] *= 10.0
        return s.to(DEV)
    s = system()
    assert T.static_step_ok(s)
    rays = torch.from_numpy(O.synthetic_rays(16, 16)).to(DEV)
    rays[:, 6], rays[:, 7] = 2.5, 5.5
    N = rays.shape[0]
    target = torch.rand(N, 3, generator=torch.Generator().manual_seed(2)).to(DEV)

    def run(gt, static):
        s.zero_grad(set_to_none=True)
        ex = {"mirror_mask": gt, "is_eval": False, "train_geometry_stage": False}
        if static:
            ex.update(_static=True, _gt_valid=True)
        res = s(rays, ex)
        loss = ((res["rgb_fine"] - target) ** 2).mean() + ((res["rgb_coarse"] - target) ** 2).mean() \
            + 0.05 * ((res["mirror_mask_fine"] - gt) ** 2).mean()
        loss.backward()
        return res, float(loss), {k: (None if v.grad is None else v.grad.clone()) for k, v in s.named_parameters()}

    for name, gt in (("a third", (torch.arange(N, device=DEV) % 3 == 0).float()), ("none", torch.zeros(N, device=DEV)),
                     ("all", torch.ones(N, device=DEV))):
        res_h, loss_h, g_h = run(gt, False)
        res_s, loss_s, g_s = run(gt, True)
        for k in ("rgb_coarse", "rgb_fine", "depth_fine", "mirror_mask_fine", "opacity_fine", "x_surface_fine"):
            assert torch.equal(res_h[k], res_s[k]), (name, k)
        assert loss_h == loss_s, name
        for k in g_h:
            assert (g_h[k] is None) == (g_s[k] is None), (name, k)
            if g_h[k] is None:
                continue
            scale = float(g_h[k].abs().max())
            err = float((g_h[k] - g_s[k]).abs().max())
            # (table: the hashed levels accumulate in fixed point -- exact integer sums -- the coarse levels in fp32 private copies,
            #  whose atomics arrive in another order when the launch is sized for the capacity)
            assert err <= (1e-6 if k.endswith("encoder.embeddings") else 2e-5) * scale + 1e-12, (name, k, err, scale)
    # the whole step: train_step(gt_valid=True) takes the static route for this model as well
    a, b = system(), system()
    oa = torch.optim.Adam(list(a.parameters()), lr=1e-3, fused=True)
    ob = torch.optim.Adam(list(b.parameters()), lr=1e-3, fused=True)
    gt = (torch.arange(N, device=DEV) % 3 == 0).float()
    for _ in range(2):
        la = T.train_step(a, oa, rays, target, gt, gt_valid=True)
        lb = T.train_step(b, ob, rays, target, gt)
        assert abs(float(la) - float(lb)) <= 1e-6 * max(1.0, abs(float(lb)))
    for (k, qa), qb in zip(a.named_parameters(), b.parameters()):
        assert float((qa - qb).abs().max()) <= 2 * 1e-3 + 1e-6, k


def test_captured_loop_crosses_scale_lowering_recapture_and_fp32_pin():
    """ONE run through every state of the guard machine around the captured step.  Trained weights (fixture G11),
    TotalLoss, batches of the analytic scene named by integers (tests/golden/make_golden_spike.py): batch 284's backward outgrows the
    f16 range at the default gradient scale -> its update is skipped on the device, the scale is lowered by 2^4 and the step is
    captured AGAIN (the model stays on the split arithmetic); later a batch whose rays start 100 units away trips the encoding range
    -> nothing to adapt: that update is skipped too, the models are pinned to fp32 and the loop continues on train_step.  At the end:
    one capture per gradient scale used (two or three), finite weights that moved, the skipped updates counted."""
    import sys
    from types import SimpleNamespace
    import numpy as np
    import warnings
    import mirror_nerf_amd as M
    from mirror_nerf_amd import training as T
    from mirror_nerf_amd.mirror_nerf import precision_of
    from tests.golden import fixtures as FX
    sys.path.insert(0, FX.HERE)
    import make_golden_trained as SC
    system = M.NeRFSystem(T.default_hparams(N_importance=64, perturb=0.0, noise_std=0.0))
    z = np.load(f"{FX.HERE}/g11_trained_weights.npz")
    for name, mod in (("coarse", system.nerf_coarse), ("fine", system.nerf_fine)):
        mod.load_state_dict({k[len(name) + 2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(name + "__")})
    system.to(DEV)
    rays, rgbs, masks = (torch.from_numpy(a).to(DEV) for a in SC.scene_views(48, 100, 100))
    opt = T.FlatAdam(list(system.models.values()), lr=1e-5)
    loss_fn = T.total_loss_fn(SimpleNamespace(model_type="nerf"), epoch=5)
    step = T.GraphedTrainStep(system, opt, 1024, loss_fn=loss_fn, epoch=5, gt_valid=True)
    captures = []
    real_capture = step.capture
    step.capture = lambda: (captures.append(len(captures)), real_capture())[1]
    w0 = [fp.detach().clone() for fp in opt.flats]
    far_at = 40
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        for i in range(60):
            idx = torch.from_numpy(np.random.RandomState(284 if i == 0 else 1000 + i).randint(rays.shape[0], size=1024)).to(DEV)
            r = rays[idx].contiguous()
            if i == far_at:
                r[:, :3] += 100.0
            loss = step(r, rgbs[idx].contiguous(), masks[idx].contiguous())
            if i in (5, far_at + 5, 59):
                torch.cuda.synchronize()
                assert bool(torch.isfinite(loss).all()), i
            if i == far_at - 1:
                torch.cuda.synchronize()
                reds = [m.__dict__.get("_mnrf_seed_reduction", 0) for m in system.models.values()]
                # a new capture per adaptation event: the fine model after batch 284, the coarse model when a later batch outgrows
                # ITS scale (each model has its own two-rung ladder; two models adapting at one settle share a capture)
                n_adapt = sum(r // 4 for r in reds)
                assert not step.ended and max(reds) in (4, 8) and 2 <= len(captures) <= 1 + n_adapt, (step.ended, reds, captures)
                assert all(precision_of(m) == "split" for m in system.models.values())
                n_captures_before_pin = len(captures)
    torch.cuda.synchronize()
    msgs = [str(w.message) for w in caught]
    assert any("gradient scale is lowered" in m for m in msgs), msgs[:5]
    assert any("captured training step" in m and "fp32" in m for m in msgs), msgs[:5]
    assert step.ended and len(captures) == n_captures_before_pin
    assert all(precision_of(m) == "fp32" for m in system.models.values())
    skipped = sum(int(t.item()) for t in opt._skipped)
    assert skipped >= 2, skipped                      # (the tripping replays; a replay queued before its trip was read trips as well)
    for fp, w in zip(opt.flats, w0):
        assert bool(torch.isfinite(fp).all()) and float((fp.detach() - w).abs().max()) > 0.0
    assert opt._calls == 60
