#!/usr/bin/env python3
"""Fixtures G11: parity on TRAINED weights (the numerical hazards bite "mainly with trained checkpoints whose
high-frequency bands carry weight"; every other fixture uses seeded random-init weights).

Build-container only (imports /root/reference read-only through `_ref_import`):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_trained.py train [steps]    # ~1 h on 8 cores; writes g11_trained_weights.npz
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_trained.py capture          # writes g11_*.npz from those weights

`train` optimises the REFERENCE (`train.NeRFSystem.forward` + `losses.TotalLoss`, Adam, the reference's training
defaults: perturb = noise_std = 1, batches of random rays over all views) on an analytic mirror scene rendered by the
ray tracer below: a checkered floor, a textured sphere and a rectangular plane mirror, 24 views of 64x64, ground-truth
colour WITH the mirror reflection and ground-truth mirror mask.  No dataset ships with the reference and none can be
downloaded here; the scene only has to make the optimiser pull the weights away from their initial distribution
(sharp texture edges -> high-frequency encoding columns, opaque surfaces -> large densities).
`capture` then records, from the reference on those weights: render_rays (64 + 128, test and train mode),
eval.batched_inference with two bounces, the gradients of a train step, and a 48x48 novel view with its analytic ground
truth (PSNR of the reference render; the HIP render has to land within 0.1 dB of it, BASELINE north star).
A second weight set `g11_rough_weights` is SYNTHESISED from the trained one (the 2^7..2^9 frequency columns of
xyz_encoding_1/5 scaled up, density head scaled so that sigma reaches ~1e3): the worst case for the split-f16 arithmetic
that a long training run could produce, used by the range-guard tests.
"""
import json
import os
import sys
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.dont_write_bytecode = True

F32 = np.float32

# ----------------------------------------------------------------------------- analytic scene
MIRROR_Y = 1.2                       # mirror in the plane y = MIRROR_Y, facing -y
MIRROR_X = (-1.3, 1.3)
MIRROR_Z = (0.15, 1.7)
SPHERE_C = np.array([0.45, 0.1, 0.55])
SPHERE_R = 0.55
FLOOR_HALF = 3.0
SKY = np.array([0.55, 0.7, 0.9])


def _floor_color(p):
    c = (np.floor(p[:, 0] * 2.5) + np.floor(p[:, 1] * 2.5)) % 2
    a = np.array([0.85, 0.82, 0.75])
    b = np.array([0.15, 0.2, 0.3])
    return np.where(c[:, None] > 0, a, b)


def _sphere_color(p):
    n = (p - SPHERE_C) / SPHERE_R
    s = 0.5 + 0.5 * np.sin(9.0 * n[:, 2:3] + 4.0 * np.arctan2(n[:, 1:2], n[:, 0:1]))
    return np.concatenate([0.9 * s, 0.25 + 0.5 * (1 - s), 0.3 + 0.2 * s], 1)


def trace(o, d, bounce=1):
    """Nearest hit among floor, sphere and mirror; mirror hits are followed once.  Returns (rgb (N,3), mirror (N,) bool)."""
    o, d = np.asarray(o, np.float64), np.asarray(d, np.float64)
    n = o.shape[0]
    t_best = np.full(n, np.inf)
    kind = np.zeros(n, np.int64)                      # 0 sky, 1 floor, 2 sphere, 3 mirror
    with np.errstate(divide="ignore", invalid="ignore"):
        t = -o[:, 2] / d[:, 2]                        # floor z = 0
        p = o + d * t[:, None]
        ok = (t > 1e-6) & (np.abs(p[:, 0]) < FLOOR_HALF) & (np.abs(p[:, 1]) < FLOOR_HALF)
        t_best, kind = np.where(ok & (t < t_best), t, t_best), np.where(ok & (t < t_best), 1, kind)
        oc = o - SPHERE_C
        b = (oc * d).sum(1)
        c = (oc * oc).sum(1) - SPHERE_R ** 2
        disc = b * b - c
        t = -b - np.sqrt(np.maximum(disc, 0))
        ok = (disc > 0) & (t > 1e-6)
        upd = ok & (t < t_best)
        t_best, kind = np.where(upd, t, t_best), np.where(upd, 2, kind)
        t = (MIRROR_Y - o[:, 1]) / d[:, 1]
        p = o + d * t[:, None]
        ok = (t > 1e-6) & (d[:, 1] > 0) & (p[:, 0] > MIRROR_X[0]) & (p[:, 0] < MIRROR_X[1]) & (p[:, 2] > MIRROR_Z[0]) & (p[:, 2] < MIRROR_Z[1])
        upd = ok & (t < t_best)
        t_best, kind = np.where(upd, t, t_best), np.where(upd, 3, kind)
    rgb = np.tile(SKY, (n, 1))
    hit = o + d * np.where(np.isfinite(t_best), t_best, 0.0)[:, None]
    m = kind == 1
    rgb[m] = _floor_color(hit[m])
    m = kind == 2
    rgb[m] = _sphere_color(hit[m])
    m = kind == 3
    if m.any():
        if bounce > 0:
            dr = d[m].copy()
            dr[:, 1] = -dr[:, 1]
            rgb[m] = trace(hit[m], dr, bounce - 1)[0]
        else:
            rgb[m] = 0.5
    return rgb.astype(F32), kind == 3


def scene_views(n_views, H, W, held_out=False):
    from oracle import mirror_nerf_oracle as O
    rays, rgbs, masks = [], [], []
    for v in range(n_views):
        a = (-0.9 + 1.8 * (v + (0.5 if held_out else 0.0)) / max(1, n_views - (0 if held_out else 1))) if n_views > 1 else 0.2
        eye = (2.6 * np.sin(a), -2.6 * np.cos(a) + 0.2, 0.9 + 0.5 * np.cos(2.3 * v))
        pose = O.look_at_pose(eye=eye, target=(0.1, 0.6, 0.6))
        focal = 0.5 * W / np.tan(0.5 * 0.9)
        o, d = O.get_rays(O.get_ray_directions(H, W, focal), pose)
        r = np.concatenate([o, d, np.full((o.shape[0], 1), 0.05, F32), np.full((o.shape[0], 1), 8.0, F32)], 1).astype(F32)
        c, m = trace(o, d)
        rays.append(r), rgbs.append(c), masks.append(m.astype(F32))
    return np.concatenate(rays), np.concatenate(rgbs), np.concatenate(masks)


# ----------------------------------------------------------------------------- reference training
def ref_system(n_importance=64):
    import _ref_import as R
    R.install()
    import torch
    import train as ref_train
    hp = R.get_hparams(predict_normal=True, predict_mirror_mask=True, trace_secondary_rays=True, N_samples=64,
                       N_importance=n_importance, perturb=1.0, noise_std=1.0, only_trace_rays_in_mirrors=True,
                       max_recursive_level=1)
    torch.manual_seed(0)
    system = ref_train.NeRFSystem(hp)
    system.train_dataset = types.SimpleNamespace(white_back=False)
    system.train_geometry_stage = False
    return system, hp


def train(steps, batch=512):
    import torch
    torch.set_num_threads(int(os.environ.get("MNRF_THREADS", "8")))
    system, hp = ref_system()
    rays, rgbs, masks = scene_views(24, 64, 64)
    print(f"scene: {rays.shape[0]} rays, {masks.mean()*100:.1f} % mirror pixels")
    rays_t, rgbs_t, masks_t = (torch.from_numpy(x) for x in (rays, rgbs, masks))
    opt = torch.optim.Adam(system.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(1)
    epoch = hp.train_geometry_stage_end_epoch + 1
    t0 = time.time()
    out = os.path.join(HERE, "g11_trained_weights.npz")
    for it in range(steps):
        idx = torch.randint(0, rays_t.shape[0], (batch,), generator=g)
        b = {"rays": rays_t[idx], "rgbs": rgbs_t[idx], "mirror_mask": masks_t[idx]}
        extra = {"is_eval": False, "mirror_mask": b["mirror_mask"], "only_one_field": False, "only_one_field_fine_epoch": 2,
                 "current_epoch": epoch, "train_geometry_stage": False}
        res = system(b["rays"], extra)
        loss, parts = system.loss(res, b, False, epoch)
        opt.zero_grad()
        loss.backward()
        opt.step()
        if it % 10 == 0 or it == steps - 1:
            with torch.no_grad():
                mse = float(((res["rgb_fine"] - b["rgbs"]) ** 2).mean())
            print(f"step {it:4d}  loss {loss.item():.4f}  psnr {-10*np.log10(mse):.2f}  "
                  + " ".join(f"{k[:-5]} {float(v):.4f}" for k, v in parts.items()) + f"  [{time.time()-t0:.0f} s]", flush=True)
        if it % 50 == 49 or it == steps - 1:
            arrs = {}
            for mname, mod in (("coarse", system.nerf_coarse), ("fine", system.nerf_fine)):
                for k, v in mod.state_dict().items():
                    arrs[f"{mname}__{k}"] = v.detach().numpy().copy()
            arrs["meta"] = np.array(json.dumps(dict(steps=it + 1, batch=batch, lr=1e-3, loss=float(loss.item()), views=24, res=64)))
            np.savez_compressed(out, **arrs)
    print("wrote", out, f"{os.path.getsize(out)/1e6:.1f} MB")


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "train"
    if cmd == "train":
        train(int(sys.argv[2]) if len(sys.argv) > 2 else 400)
    elif cmd == "views":
        r, c, m = scene_views(2, 64, 64)
        print(r.shape, c.mean(0), m.mean())
    else:
        import make_golden_trained_capture as C
        C.main()
