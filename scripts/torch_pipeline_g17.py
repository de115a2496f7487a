"""CPU, build container or anywhere: the train step of fixture g17_tcnn_train_grads as a plain-torch pipeline (tests/torch_ref.py
nodes + the oracle's deterministic sampling), against the reference's gradients stored in the fixture."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import mirror_nerf_oracle as O
from tests import torch_ref as R
from tests.golden import fixtures as FX
from tests.golden import make_golden_loss as GL

fx = FX.Fixture(sys.argv[1] if len(sys.argv) > 1 else "g17_tcnn_train_grads")
loss_fn = getattr(GL, fx.meta["loss"])
W = []
for prefix, which in (("coarse__", 0), ("fine__", 1)):
    w = FX.tcnn_weights(fx, prefix, which)
    cfg = w.pop("_cfg")
    W.append({k: torch.from_numpy(v).requires_grad_(True) for k, v in w.items()})
t = lambda k: torch.from_numpy(fx.inputs[k].copy())
Z = {}
import os
ZIN = np.load(os.environ['ZIN']) if os.environ.get('ZIN') else None
ZLV = os.environ.get('ZLV', 'both')


def render(rays, tag):
    N = rays.shape[0]
    zs = torch.from_numpy(O.torch_linspace(0, 1, 64))
    z = (rays[:, 6:7] * (1 - zs) + rays[:, 7:8] * zs).detach()
    out = {}
    for typ, w in (("coarse", W[0]), ("fine", W[1])):
        if typ == "fine":
            mid = 0.5 * (z[:, :-1] + z[:, 1:])
            znew = O.sample_pdf(mid.numpy(), out["weights_coarse"][:, 1:-1].detach().numpy(), 64, det=True)
            z = torch.from_numpy(np.sort(np.concatenate([z.numpy(), znew], -1), -1).astype(np.float32))
            if ZIN is not None and ZLV in ("both", tag):
                zin = torch.from_numpy(ZIN[f"{tag}_fine"])
                if os.environ.get("ZROW"):
                    rows = [int(v) for v in os.environ["ZROW"].split(",")]
                    z = z.clone()
                    z[rows] = zin[rows]
                else:
                    eps = float(os.environ.get("ZEPS", 1))
                    z = (z.double() + eps * (zin.double() - z.double())).float()
        S = z.shape[1]
        xyz = (rays[:, None, :3] + rays[:, None, 3:6] * z[..., None]).reshape(-1, 3)
        dd = rays[:, None, 3:6].expand(N, S, 3).reshape(-1, 3)
        sigma, rgb, pn, mir = R.tcnn_field(w, torch.cat([xyz, dd], 1), cfg)
        c = R.composite(rays, sigma.view(N, S), z, None, rgb.view(N, S, 3), mir.view(N, S), pn.view(N, S, 3), None)
        for k, v in (("weights", "weights"), ("opacity", "opacity"), ("rgb", "rgb"), ("depth", "depth"), ("mirror_mask", "mask"),
                     ("surface_normal", "sn"), ("x_surface", "xs")):
            out[f"{k}_{typ}"] = c[v]
        Z[f"{tag}_{typ}"] = z
    return out


rays, gt = t("rays"), t("gt_mask")
r = render(rays, "l0")
sec = R.reflect(rays, r["x_surface_fine"], r["surface_normal_fine"], gt, True)
r2 = render(sec, "l1")
for typ in ("coarse", "fine"):
    r[f"rgb_{typ}"] = R.blend(r[f"rgb_{typ}"], r2[f"rgb_{typ}"], gt, True)
loss = loss_fn(r, t("target"), gt)
loss.backward()
print(f"loss {loss.item():.6f} (reference {float(fx.outputs['loss']):.6f})")
for k in ("rgb_fine", "depth_fine", "z_vals_fine"):
    got = Z["l0_fine"] if k == "z_vals_fine" else r[k]
    print(f"  forward {k}: {np.abs(got.detach().numpy() - fx.outputs[k]).max():.2e}")
for mname, w in (("coarse", W[0]), ("fine", W[1])):
    for k, p in w.items():
        if k == "encoder.embeddings":
            lv, val, _ = FX.table_grad_summary(p.grad.numpy(), cfg, fx.outputs[f"table_idx__{mname}"])
            wl, wv = fx.outputs[f"table_levels__{mname}"], fx.outputs[f"table_val__{mname}"]
            print(f"  {mname:6s} table: level norms {np.max(np.abs(lv[:,1]-wl[:,1]))/wl[:,1].max():.2e}  entries {np.max(np.abs(val-wv))/np.abs(wv).max():.2e}")
            continue
        want = fx.outputs[f"grad__{mname}__{k}"]
        if np.abs(want).max() == 0 or p.grad is None:
            continue
        print(f"  {mname:6s} {k:24s} {np.max(np.abs(p.grad.numpy() - want)) / np.abs(want).max():.2e}")
if len(sys.argv) > 2:
    np.savez(sys.argv[2], **{k: v.numpy() for k, v in Z.items()}, sec=sec.detach().numpy())
