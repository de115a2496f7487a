#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE.

Build-container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
It imports /root/reference (read-only) through `_ref_import`, runs the reference
functions on seeded inputs and stores inputs + outputs as .npz.  It also runs the
oracle on the same inputs and prints the max-abs difference per fixture, so a
fixture is never committed that the oracle does not already reproduce.

Fixture ids follow SURVEY.md section 8(c): G1 embedding, G2 field, G3 coarse-only
render, G4 coarse+fine render variants, G5 injected-randomness render, G6 train
recursion, G7 eval recursion, G8 roughness (all-mirror).
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.dont_write_bytecode = True

import _ref_import as R  # noqa: E402

R.install()
import torch  # noqa: E402
from models.mirror_nerf import Embedding, MirrorNeRF  # noqa: E402  (reference)
from models.rendering import render_rays as ref_render_rays  # noqa: E402

import weights as W  # noqa: E402
from oracle import mirror_nerf_oracle as O  # noqa: E402

torch.set_num_threads(8)
PER_SAMPLE = ("pred_normal_", "normal_coarse", "normal_fine")


def ref_models(seed, n_models, tweaks):
    """Reference modules under `seed`; checks tests/golden/weights.py rebuilds them."""
    torch.manual_seed(seed)
    mods = [MirrorNeRF(in_channels_xyz=63, in_channels_dir=27, predict_normal=True,
                       predict_mirror_mask=True) for _ in range(n_models)]
    sds = W.make_state_dict(seed, n_models)
    for m, sd in zip(mods, sds):
        ref_sd = {k: v.detach().numpy() for k, v in m.state_dict().items()}
        assert list(ref_sd) == O.field_param_names(), "state_dict key order changed"
        for k in ref_sd:
            assert np.array_equal(ref_sd[k], sd[k]), f"seed rebuild mismatch: {k}"
        W.apply_tweaks(sd, tweaks)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        m.eval()
    return mods, sds


def pick_rays(n, seed, H=400, W_=400):
    rays = O.synthetic_rays(H, W_)
    idx = np.random.RandomState(seed).choice(rays.shape[0], n, replace=False)
    return rays[np.sort(idx)]


def to_np(d):
    return {k: v.detach().cpu().numpy() for k, v in d.items()}


def save(name, meta, inputs, outputs, keep_per_sample=True):
    arrs = {}
    for k, v in inputs.items():
        arrs["in__" + k] = np.asarray(v)
    for k, v in outputs.items():
        if not keep_per_sample and any(k.startswith(p) for p in PER_SAMPLE):
            continue
        arrs["out__" + k] = np.asarray(v)
    arrs["meta"] = np.array(json.dumps(meta))
    path = os.path.join(os.environ.get("MNRF_GOLDEN_OUT", HERE), name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"  wrote {name}.npz  {os.path.getsize(path)/1024:.0f} KiB")


def off_fraction(a, b, base=1e-4):
    """Fraction of rays (rows) on which two captures of a key differ by more than the parity bar (depth-like keys:
    relative to far = 8).  Stored per fixture for the reference's own fp32-vs-fp64 runs: with trained weights a few rays
    per hundred flip an inverse-CDF bin and move by 1e-2 while all others agree to 1e-6 -- the max-abs floor alone would
    hide how rarely that happens."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0
    d = np.abs(a - b).reshape(a.shape[0], -1).max(1)
    return float((d > base).mean())


def report(tag, ref, orc, skip=()):
    worst = 0.0
    for k, v in ref.items():
        if k in skip:
            continue
        assert k in orc, f"{tag}: oracle lacks key {k}"
        assert orc[k].shape == v.shape, (tag, k, orc[k].shape, v.shape)
        d = float(np.max(np.abs(orc[k].astype(np.float64) - v.astype(np.float64)))) if v.size else 0.0
        worst = max(worst, d)
        if d > 2e-5:
            print(f"    !! {tag}:{k} max-abs {d:.3e}")
    extra = set(orc) - set(ref)
    assert not extra, f"{tag}: oracle has extra keys {extra}"
    print(f"  {tag}: oracle vs reference worst max-abs {worst:.2e} over {len(ref)} keys")
    return worst


EMB = {"xyz": Embedding(10), "dir": Embedding(4)}
EMB_O = {"xyz": 10, "dir": 4}


# ---------------------------------------------------------------- G1
def g1():
    rs = np.random.RandomState(1)
    x = rs.uniform(-8, 8, (256, 3)).astype(np.float32)
    d = rs.normal(size=(256, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    e10 = Embedding(10)(torch.from_numpy(x)).numpy()
    e4 = Embedding(4)(torch.from_numpy(d)).numpy()
    e0 = Embedding(0)(torch.from_numpy(x)).numpy()
    report("G1", {"e10": e10, "e4": e4, "e0": e0},
           {"e10": O.embedding(x, 10), "e4": O.embedding(d, 4), "e0": O.embedding(x, 0)})
    ls = {f"linspace_{n}": torch.linspace(0, 1, n).numpy() for n in (2, 5, 64, 128, 192)}
    for n in (2, 5, 64, 128, 192):
        assert np.array_equal(ls[f"linspace_{n}"], O.torch_linspace(0, 1, n))
    save("g1_embedding", {"seed": 1}, {"x": x, "d": d}, {"e10": e10, "e4": e4, "e0": e0, **ls})


# ---------------------------------------------------------------- G2
def g2():
    mods, sds = ref_models(0, 1, W.OPAQUE)
    m, sd = mods[0], sds[0]
    rs = np.random.RandomState(2)
    xyz = rs.uniform(-3, 3, (512, 3)).astype(np.float32)
    d = rs.normal(size=(512, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    x30 = np.concatenate([xyz, O.embedding(d, 4)], 1)
    outs = {}
    with torch.no_grad():
        full = to_np(m(torch.from_numpy(x30), compute_normal=False, sigma_only=False,
                       embedding_xyz=EMB["xyz"], embedding_dir=EMB["dir"]))
        sonly = to_np(m(torch.from_numpy(xyz), compute_normal=False, sigma_only=True,
                        embedding_xyz=EMB["xyz"], embedding_dir=EMB["dir"]))
    fulln = to_np(m(torch.from_numpy(x30.copy()), compute_normal=True, sigma_only=False,
                    embedding_xyz=EMB["xyz"], embedding_dir=EMB["dir"]))
    o_full = O.field_forward(sd, x30, False, False)
    o_s = O.field_forward(sd, xyz, True, False)
    o_n = O.field_forward(sd, x30, False, True)
    report("G2 full", full, o_full)
    report("G2 sigma_only", sonly, o_s)
    report("G2 normal", fulln, o_n)
    for k, v in full.items():
        outs["full__" + k] = v if k != "geo_feat" else v[:, :8]
    for k, v in sonly.items():
        outs["sigma_only__" + k] = v if k != "geo_feat" else v[:, :8]
    outs["full__normal"] = fulln["normal"]
    save("g2_field", {"seed": 0, "tweaks": W.OPAQUE, "checksum": W.checksum(sd)},
         {"x30": x30}, outs)


# ---------------------------------------------------------------- G3/G4/G5
def render_case(name, n_rays, seed_rays, n_imp, tweaks=W.OPAQUE, use_disp=False, white_back=False,
                test_time=False, perturb=0, noise_std=0, inject=None, keep_per_sample=True,
                chunk=32768, **kw):
    mods, sds = ref_models(0, 2, tweaks)
    rays = pick_rays(n_rays, seed_rays)
    models = {"coarse": mods[0]}
    models_o = {"coarse": sds[0]}
    if n_imp > 0 and not kw.get("only_one_field", False):
        models["fine"] = mods[1]
        models_o["fine"] = sds[1]
    okw = dict(kw)
    inputs = {"rays": rays}
    patched = {}
    if inject:
        rs = np.random.RandomState(inject)
        S_f = 64 + n_imp
        draws = {"_perturb_rand": rs.uniform(size=(n_rays, 64)).astype(np.float32),
                 "_noise_coarse": rs.normal(size=(n_rays, 64)).astype(np.float32),
                 "_u": rs.uniform(size=(n_rays, n_imp)).astype(np.float32),
                 "_noise_fine": rs.normal(size=(n_rays, S_f)).astype(np.float32)}
        okw.update(draws)
        inputs.update({k[1:]: v for k, v in draws.items()})
        # the reference draws, in order: rand_like(z_vals) [perturb], randn_like(sigmas)
        # [coarse], rand(N, n_imp) [sample_pdf], randn_like(sigmas) [fine]
        q_randn = [draws["_noise_coarse"], draws["_noise_fine"]]
        for fn in ("rand_like", "randn_like", "rand"):
            patched[fn] = getattr(torch, fn)
        torch.rand_like = lambda t, **k: torch.from_numpy(draws["_perturb_rand"])
        torch.randn_like = lambda t, **k: torch.from_numpy(q_randn.pop(0))
        torch.rand = lambda *a, **k: torch.from_numpy(draws["_u"])
    try:
        grad_needed = kw.get("compute_normal", True)
        ctx = torch.enable_grad() if grad_needed else torch.no_grad()
        with ctx:
            ref = to_np(ref_render_rays(models, EMB, torch.from_numpy(rays), 64, use_disp, perturb,
                                        noise_std, n_imp, chunk, white_back, test_time, **kw))
    finally:
        for fn, f in patched.items():
            setattr(torch, fn, f)
    orc = O.render_rays(models_o, EMB_O, rays, 64, use_disp, perturb, noise_std, n_imp, chunk,
                        white_back, test_time, **okw)
    # the reference's own fp32 noise floor: the same call in float64
    floor = {}
    if not inject:
        import copy
        m64 = {k: copy.deepcopy(v).double() for k, v in models.items()}
        with ctx:
            ref64 = to_np(ref_render_rays(m64, EMB, torch.from_numpy(rays).double(), 64, use_disp,
                                          perturb, noise_std, n_imp, chunk, white_back, test_time, **kw))
        floor = {k: float(np.max(np.abs(ref64[k] - ref[k].astype(np.float64)))) for k in ref}
        floor_frac = {k: off_fraction(ref64[k], ref[k]) for k in ref}
        big = {k: f"{v:.1e}" for k, v in floor.items() if v > 2e-5}
        if big:
            print(f"    reference fp32-vs-fp64 floor > 2e-5: {big}")
            print("    fraction of rays off by more than the 1e-4 bar:", {k: round(v, 4) for k, v in floor_frac.items() if v > 0})
    # per-sample positions are not stable at 1e-4; compare them loosely here
    report(name, ref, orc, skip=("z_vals_fine", "weights_fine", "pred_normal_fine", "normal_fine"))
    meta = dict(seed=0, n_models=2, tweaks=tweaks, checksum=[W.checksum(s) for s in sds],
                N_samples=64, N_importance=n_imp, use_disp=use_disp, white_back=white_back,
                test_time=test_time, perturb=perturb, noise_std=noise_std, chunk=chunk,
                kwargs={k: v for k, v in kw.items()}, injected=bool(inject), floor=floor,
                floor_frac=floor_frac if not inject else {})
    save(name, meta, inputs, ref, keep_per_sample)


# ---------------------------------------------------------------- G6
class _HP(dict):
    __getattr__ = dict.__getitem__


def train_case(name, n_rays, gt_mode, tweaks, **hp_over):
    import train as ref_train

    hp = R.get_hparams(predict_normal=True, predict_mirror_mask=True, trace_secondary_rays=True,
                       N_samples=64, N_importance=64, perturb=0, noise_std=0, chunk=hp_over.pop("chunk", 32768),
                       **hp_over)
    torch.manual_seed(0)
    system = ref_train.NeRFSystem(hp)
    _, sds = ref_models(0, 2, tweaks)
    system.nerf_coarse.load_state_dict({k: torch.from_numpy(v) for k, v in sds[0].items()})
    system.nerf_fine.load_state_dict({k: torch.from_numpy(v) for k, v in sds[1].items()})
    system.train_dataset = types.SimpleNamespace(white_back=False)
    rays = pick_rays(n_rays, 6)
    rs = np.random.RandomState(66)
    if gt_mode == "gt25":
        gt = (rs.uniform(size=n_rays) < 0.25).astype(np.float32)
    elif gt_mode == "gt0":
        gt = np.zeros(n_rays, dtype=np.float32)          # valid GT without a single mirror pixel
    else:
        gt = -np.ones(n_rays, dtype=np.float32)  # invalid GT -> predicted mask
    is_eval = hp_over.get("is_eval", False)
    extra = {"mirror_mask": torch.from_numpy(gt.copy()), "is_eval": is_eval,
             "train_geometry_stage": False}
    ref = to_np(system(torch.from_numpy(rays), extra))
    hp_o = dict(N_samples=64, use_disp=False, perturb=0, noise_std=0, N_importance=64,
                chunk=hp.chunk, trace_secondary_rays=True, only_one_field=False,
                max_recursive_level=hp.max_recursive_level,
                only_trace_rays_in_mirrors=hp.only_trace_rays_in_mirrors, for_vis=hp.for_vis)
    orc = O.render_train({"coarse": sds[0], "fine": sds[1]}, EMB_O, rays, hp_o,
                         {"mirror_mask": gt.copy(), "is_eval": is_eval, "train_geometry_stage": False})
    report(name, ref, orc, skip=("z_vals_fine", "weights_fine", "pred_normal_fine", "normal_fine"))
    n_refl = int((ref["rgb_fine_direct"] != ref["rgb_fine"]).any(-1).sum()) if "rgb_fine_direct" in ref else 0
    print(f"    rays whose colour changed by reflection: {n_refl}/{n_rays}")
    meta = dict(seed=0, n_models=2, tweaks=tweaks, checksum=[W.checksum(s) for s in sds], hp=hp_o,
                is_eval=is_eval, gt_mode=gt_mode)
    save(name, meta, {"rays": rays, "gt_mask": gt}, ref, keep_per_sample=False)


# ---------------------------------------------------------------- G7/G8
def eval_case(name, n_rays, tweaks, max_level, chunk=32768, rough=False, n_imp=64, want_floor=False, only_mirror_rays=False):
    import eval as ref_eval

    ref_eval.dataset = types.SimpleNamespace(white_back=False)
    hp = R.get_hparams()
    args = types.SimpleNamespace(**vars(hp))
    args.predict_normal = True
    args.predict_mirror_mask = True
    args.only_one_field = False
    args.max_recursive_level = max_level
    args.app_control_mirror_roughness = rough
    args.app_reflection_substitution = False
    args.app_place_new_mirror = False
    args.app_reflect_newly_placed_objects = False
    args.trace_ray_times = 2
    args.normal_noise_std = 0.05
    mods, sds = ref_models(0, 2, tweaks)
    rays = pick_rays(n_rays, 7)
    if only_mirror_rays:
        # roughness at level 0 needs an all-mirror chunk (the reference adds M = sum(mask) rows to N rows):
        # keep the rays the level-0 render calls mirror; their reflections at level 1 are mirror only in part
        with torch.no_grad():
            r0 = ref_render_rays({"coarse": mods[0], "fine": mods[1]}, EMB, torch.from_numpy(pick_rays(4 * n_rays, 7)), 64, False, 0, 0,
                                 n_imp, chunk, False, test_time=True, compute_normal=False)
        rays = pick_rays(4 * n_rays, 7)[(r0["mirror_mask_fine"] > 0.502).numpy()][:n_rays]
        print(f"    kept {rays.shape[0]} level-0 mirror rays")
    draws = []
    orig = torch.randn_like
    rs = np.random.RandomState(88)

    def fake_randn_like(t, **k):
        if t.dim() == 2 and t.shape[-1] == 3:
            a = rs.normal(size=tuple(t.shape)).astype(np.float32)
            draws.append(a)
            return torch.from_numpy(a).to(t.dtype)
        return torch.zeros_like(t)

    torch.randn_like = fake_randn_like
    try:
        ref = to_np(ref_eval.batched_inference(
            {"coarse": mods[0], "fine": mods[1]}, EMB, torch.from_numpy(rays), 64, n_imp, False,
            chunk, args=args, trace_secondary_rays=True, normal_noise_std=args.normal_noise_std))
    finally:
        torch.randn_like = orig
    args_o = dict(predict_normal=True, only_one_field=False, only_one_field_fine_epoch=2,
                  max_recursive_level=max_level, app_control_mirror_roughness=rough,
                  trace_ray_times=2, normal_noise_std=args.normal_noise_std)
    orc = O.render_eval({"coarse": sds[0], "fine": sds[1]}, EMB_O, rays, 64, n_imp, False, chunk,
                        args_o, normal_noise=iter(draws))
    report(name, ref, orc, skip=("z_vals_fine", "weights_fine", "pred_normal_fine"))
    print(f"    mirror rays at level 0: {int((ref['mirror_mask_fine'] > 0.5).sum())}/{n_rays}; "
          f"normal-noise draws: {len(draws)}")
    floor, floor_frac = {}, {}
    if want_floor and not rough:      # the reference's own noise: the same call in float64
        import copy
        m64 = {k: copy.deepcopy(v).double() for k, v in (("coarse", mods[0]), ("fine", mods[1]))}
        ref64 = to_np(ref_eval.batched_inference(m64, EMB, torch.from_numpy(rays).double(), 64, n_imp, False, chunk, args=args,
                                                 trace_secondary_rays=True, normal_noise_std=args.normal_noise_std))
        floor = {k: float(np.max(np.abs(ref64[k] - ref[k].astype(np.float64)))) for k in ref if ref64[k].shape == ref[k].shape}
        floor_frac = {k: off_fraction(ref64[k], ref[k]) for k in ref if ref64[k].shape == ref[k].shape}
        print("    reference fp32-vs-fp64 floor:", {k: f"{v:.1e}" for k, v in floor.items() if v > 2e-5})
        print("    fraction of rays off by more than 1e-4:", {k: round(v, 4) for k, v in floor_frac.items() if v > 0})
    meta = dict(seed=0, n_models=2, tweaks=tweaks, checksum=[W.checksum(s) for s in sds],
                args=args_o, N_samples=64, N_importance=n_imp, chunk=chunk, floor=floor, floor_frac=floor_frac)
    inputs = {"rays": rays}
    for i, a in enumerate(draws):
        inputs[f"normal_noise_{i}"] = a
    save(name, meta, inputs, ref, keep_per_sample=False)


# ---------------------------------------------------------------- G9
from make_golden_loss import first_order_loss, full_loss, grad_summary  # noqa: E402


# whole gradient tensors stored by g9_train_grads_full: first trunk layer, the skip layer (319 inputs), the density head, the
# colour branch's first layer -- of both models (~1 MB)
FULL_TENSORS = ("xyz_encoding_1.0.weight", "xyz_encoding_5.0.weight", "sigma.weight", "dir_encoding.0.weight")


def grad_case(name, n_rays, loss_fn=None, full_tensors=False):
    loss_fn = loss_fn or first_order_loss
    import train as ref_train

    hp = R.get_hparams(predict_normal=True, predict_mirror_mask=True, trace_secondary_rays=True,
                       N_samples=64, N_importance=64, perturb=0, noise_std=0, only_trace_rays_in_mirrors=True,
                       max_recursive_level=1)
    torch.manual_seed(0)
    system = ref_train.NeRFSystem(hp)
    _, sds = ref_models(0, 2, W.OPAQUE)
    system.nerf_coarse.load_state_dict({k: torch.from_numpy(v) for k, v in sds[0].items()})
    system.nerf_fine.load_state_dict({k: torch.from_numpy(v) for k, v in sds[1].items()})
    system.train_dataset = types.SimpleNamespace(white_back=False)
    rays = pick_rays(n_rays, 9)
    rs = np.random.RandomState(99)
    gt = (rs.uniform(size=n_rays) < 0.3).astype(np.float32)
    target = rs.uniform(size=(n_rays, 3)).astype(np.float32)
    extra = {"mirror_mask": torch.from_numpy(gt.copy()), "is_eval": False, "train_geometry_stage": False}
    res = system(torch.from_numpy(rays), extra)
    loss = loss_fn(res, torch.from_numpy(target), torch.from_numpy(gt))
    loss.backward()
    outs = {"loss": np.array(loss.item())}
    for mname, mod in (("coarse", system.nerf_coarse), ("fine", system.nerf_fine)):
        for pn_, p_ in mod.named_parameters():
            outs[f"grad__{mname}__{pn_}"] = grad_summary(p_.grad, p_)
            if full_tensors and pn_ in FULL_TENSORS:      # entry-wise pin of whole tensors, not the 48-entry digest
                outs[f"gradfull__{mname}__{pn_}"] = (p_.grad if p_.grad is not None else torch.zeros_like(p_)).detach().numpy().copy()
    # the reference's own noise floor: the same step in float64
    g32 = {f"{mn}.{pn_}": (p_.grad.clone() if p_.grad is not None else torch.zeros_like(p_))
           for mn, mod in (("coarse", system.nerf_coarse), ("fine", system.nerf_fine)) for pn_, p_ in mod.named_parameters()}
    system.double()
    system.zero_grad()
    extra64 = {"mirror_mask": torch.from_numpy(gt.copy()).double(), "is_eval": False, "train_geometry_stage": False}
    res64 = system(torch.from_numpy(rays).double(), extra64)
    loss_fn(res64, torch.from_numpy(target).double(), torch.from_numpy(gt).double()).backward()
    floor = 0.0
    for mn, mod in (("coarse", system.nerf_coarse), ("fine", system.nerf_fine)):
        for pn_, p_ in mod.named_parameters():
            g64 = p_.grad if p_.grad is not None else torch.zeros_like(p_)
            a = g32[f"{mn}.{pn_}"].double()
            if g64.abs().max() > 0:
                floor = max(floor, float((a - g64).abs().max() / g64.abs().max()))
            if full_tensors and pn_ in FULL_TENSORS:      # the same tensors from the float64 run: the truth both fp32 sides are measured against
                outs[f"gradfull64__{mn}__{pn_}"] = g64.detach().numpy().astype(np.float32)
    print(f"  reference fp32-vs-fp64 gradient floor (relative to each tensor's max): {floor:.2e}")
    hp_o = dict(N_samples=64, use_disp=False, perturb=0, noise_std=0, N_importance=64, chunk=hp.chunk,
                trace_secondary_rays=True, only_one_field=False, max_recursive_level=1,
                only_trace_rays_in_mirrors=True, for_vis=False)
    print(f"  {name}: loss {loss.item():.6f}, |grad| coarse L1 weight {outs['grad__coarse__xyz_encoding_1.0.weight'][1]:.3e}")
    meta = dict(seed=0, n_models=2, tweaks=W.OPAQUE, checksum=[W.checksum(s) for s in sds], hp=hp_o, grad_floor=floor, loss=loss_fn.__name__)
    save(name, meta, {"rays": rays, "gt_mask": gt, "target": target}, outs)


def main():
    only = sys.argv[1:]

    def want(tag):
        return not only or any(tag.startswith(o) for o in only)

    if want("g1"):
        g1()
    if want("g2"):
        g2()
    if want("g3"):
        render_case("g3_coarse64_train", 256, 3, 0, compute_normal=True)
        render_case("g3_coarse64_test", 256, 3, 0, test_time=True, compute_normal=False,
                    keep_per_sample=False)
    if want("g4"):
        render_case("g4_fine_train", 64, 4, 128, compute_normal=True)
        render_case("g4_fine_test", 128, 4, 128, test_time=True, compute_normal=False,
                    keep_per_sample=False)
        render_case("g4_fine_disp_white", 64, 4, 128, use_disp=True, white_back=True,
                    compute_normal=False, keep_per_sample=False)
        render_case("g4_onefield_ep0", 64, 4, 128, compute_normal=False, only_one_field=True,
                    current_epoch=0, keep_per_sample=False)
        render_case("g4_onefield_ep3", 64, 4, 128, compute_normal=False, only_one_field=True,
                    current_epoch=3, keep_per_sample=False)
        render_case("g4_fine_rawinit", 64, 4, 128, tweaks=[], compute_normal=True,
                    keep_per_sample=False)
    if want("g5"):
        render_case("g5_perturb_noise", 64, 5, 64, perturb=1.0, noise_std=1.0, inject=55,
                    compute_normal=True, keep_per_sample=False)
    if want("g6"):
        train_case("g6_train_gt_compact", 96, "gt25", W.OPAQUE, only_trace_rays_in_mirrors=True,
                   max_recursive_level=1)
        train_case("g6_train_gt_full_eval", 96, "gt25", W.OPAQUE, only_trace_rays_in_mirrors=False,
                   max_recursive_level=2, is_eval=True)
        train_case("g6_train_pred_straddle", 96, "invalid", W.STRADDLE, only_trace_rays_in_mirrors=True,
                   max_recursive_level=2, is_eval=True, chunk=32)
        train_case("g6_train_nomirror_eval", 48, "invalid", W.OPAQUE, only_trace_rays_in_mirrors=True,
                   max_recursive_level=1, is_eval=True)
    if want("g6_train_forvis"):
        # for_vis: trace although no pixel is a mirror (train.py:172-178), every ray, so that the *_reflect maps exist
        train_case("g6_train_forvis_eval", 48, "gt0", W.OPAQUE, only_trace_rays_in_mirrors=False, max_recursive_level=1,
                   is_eval=True, for_vis=True)
    if want("g7"):
        eval_case("g7_eval_l1", 96, W.STRADDLE, 1)
        eval_case("g7_eval_l2_chunk32", 96, W.STRADDLE, 2, chunk=32)
    if want("g8"):
        eval_case("g8_rough_allmirror", 48, W.ALL_MIRROR, 1, rough=True)
    if want("g8b"):
        # config 4 as BASELINE names it: two bounces + roughness, with partly-mirror chunks at level 1
        eval_case("g8b_rough_l2_partial", 40, W.STRADDLE, 2, rough=True, only_mirror_rays=True)
    if want("g9"):
        grad_case("g9_train_grads", 64)
        grad_case("g9_train_grads_full", 64, full_loss, full_tensors=True)


if __name__ == "__main__":
    main()
