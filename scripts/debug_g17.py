"""Localise a G17 gradient mismatch: the train step with (a) the HIP field backward, (b) the field replaced by torch autograd
through tests/torch_ref.tcnn_field (every other node the HIP autograd functions), both against the reference's gradients."""
import sys
from types import SimpleNamespace
import numpy as np
import torch
sys.path.insert(0, ".")
import mirror_nerf_amd as M
from mirror_nerf_amd import mirror_nerf_tcnn as T
from tests import torch_ref as R
from tests.golden import fixtures as FX
from tests.golden import make_golden_loss as GL
DEV = "cuda:0"
fx = FX.Fixture(sys.argv[1] if len(sys.argv) > 1 else "g17_tcnn_train_grads")
loss_fn = getattr(GL, fx.meta["loss"])
hp = dict(fx.meta["hp"])
hp.update(model_type="nerf_tcnn", bound=fx.meta["table"]["bound"], predict_normal=True, predict_mirror_mask=True)
for k, v in (a.split("=") for a in sys.argv[2:]):
    hp[k] = eval(v)
system = M.NeRFSystem(SimpleNamespace(**hp))
for mod, (prefix, which) in ((system.nerf_coarse, ("coarse__", 0)), (system.nerf_fine, ("fine__", 1))):
    w = FX.tcnn_weights(fx, prefix, which)
    cfg = w.pop("_cfg")
    mod.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
system.to(DEV)
t = lambda k: torch.from_numpy(fx.inputs[k]).to(DEV)


def run(tag):
    system.zero_grad()
    res = system(t("rays"), {"mirror_mask": t("gt_mask"), "is_eval": False, "train_geometry_stage": False})
    loss = loss_fn(res, t("target"), t("gt_mask"))
    loss.backward()
    print(f"== {tag}: loss {loss.item():.6f} (reference {float(fx.outputs['loss']):.6f})")
    out = {}
    for mname, mod in (("coarse", system.nerf_coarse), ("fine", system.nerf_fine)):
        for pn_, p_ in mod.named_parameters():
            g = p_.grad.cpu().numpy() if p_.grad is not None else None
            out[(mname, pn_)] = g
            if pn_ == "encoder.embeddings":
                lv, val, _ = FX.table_grad_summary(g, cfg, fx.outputs[f"table_idx__{mname}"])
                wl, wv = fx.outputs[f"table_levels__{mname}"], fx.outputs[f"table_val__{mname}"]
                print(f"  {mname:6s} table: level norms {np.max(np.abs(lv[:,1]-wl[:,1]))/wl[:,1].max():.2e}  entries {np.max(np.abs(val-wv))/np.abs(wv).max():.2e}")
                continue
            want = fx.outputs[f"grad__{mname}__{pn_}"]
            if np.abs(want).max() == 0:
                continue
            print(f"  {mname:6s} {pn_:24s} {np.max(np.abs(g - want)) / np.abs(want).max():.2e}")
    return out


a = run("HIP field backward")


class TorchField:
    @staticmethod
    def apply(module, spr, xyz6, rays_, z, dirs, want_normal, table, *params):
        n = rays_.shape[0]
        xyz = (rays_[:, None, :3] + rays_[:, None, 3:6] * z[..., None]).reshape(-1, 3)
        dd = (dirs if dirs is not None else rays_[:, 3:6])[:, None, :3].expand(n, spr, 3).reshape(-1, 3)
        w = dict(module.named_parameters())
        sigma, rgb, pn, mir = R.tcnn_field(w, torch.cat([xyz, dd], 1), module.cfg)
        wn = want_normal[0] if isinstance(want_normal, tuple) else want_normal
        nrm = module.field(n * spr, rays=rays_.detach().contiguous(), z_vals=z.contiguous(), spr=spr,
                           dirs=None if dirs is None else dirs.detach().contiguous(), grad_normal=True)["normal"] \
            if wn else torch.empty(0, 3, device=rays_.device)
        return sigma, rgb, pn, mir, nrm, None


T.TcnnFieldFn = TorchField
b = run("torch-autograd field, HIP everything else")
for k in a:
    if a[k] is not None and b[k] is not None and np.abs(b[k]).max() > 0:
        print(f"  HIP vs torch field {k[0]:6s} {k[1]:24s} {np.max(np.abs(a[k]-b[k]))/np.abs(b[k]).max():.2e}")

# ---- (c) additionally replace the compositing / reflection / blend nodes by torch autograd (tests/torch_ref.py)
from mirror_nerf_amd import autograd as AG
from mirror_nerf_amd import rendering as RD


class TorchComposite:
    @staticmethod
    def apply(rays, sigma, z, noise, rgb, is_mirror, pn, nrm, white_back, detach=0, keep_mirror=None):
        N, S = z.shape
        o = R.composite(rays, sigma, z, noise, rgb.view(N, S, 3), is_mirror.view(N, S), pn.view(N, S, 3),
                        None if nrm is None else nrm.view(N, S, 3), white_back)
        return o["weights"], o["opacity"], o["rgb"], o["depth"], o["mask"], o["sn"], o.get("sng"), o.get("nd"), o["xs"]


class TorchReflect:
    @staticmethod
    def apply(rays, xs, normal, mask, compact):
        sec = R.reflect(rays, xs, normal, mask, compact)
        index = torch.nonzero(mask != 0)[:, 0].int() if compact else torch.empty(0, dtype=torch.int32, device=rays.device)
        return sec, index, sec[:, 3:6].detach()


class TorchBlend:
    @staticmethod
    def apply(base, sec, idx, mask, compact, detach_sec=False):
        if base.dim() == 1:
            return R.blend(base[:, None], sec[:, None], mask, compact)[:, 0]
        return R.blend(base, sec, mask, compact)


HIPC, HIPR, HIPB = RD.CompositeFn, AG.ReflectFn, AG.BlendFn
for tag, (c_, r_, b_) in {"+ torch composite": (TorchComposite, HIPR, HIPB), "+ torch reflect": (HIPC, TorchReflect, HIPB),
                          "+ torch blend": (HIPC, HIPR, TorchBlend), "all torch": (TorchComposite, TorchReflect, TorchBlend)}.items():
    RD.CompositeFn, AG.ReflectFn, AG.BlendFn = c_, r_, b_
    run("torch field " + tag)
