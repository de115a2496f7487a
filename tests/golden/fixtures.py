"""Loader for the committed golden vectors (tests/golden/*.npz).

Each file holds `in__*` arrays (inputs), `out__*` arrays (outputs captured from the
reference by make_golden.py) and a JSON `meta` string (seed, weight tweaks,
checksum, call arguments, the reference's own fp32-vs-fp64 noise floor)."""
import json
import os

import numpy as np

from . import weights as W

HERE = os.path.dirname(os.path.abspath(__file__))


class Fixture:
    def __init__(self, name):
        z = np.load(os.path.join(HERE, name + ".npz"))
        self.name = name
        self.meta = json.loads(str(z["meta"]))
        self.inputs = {k[4:]: z[k] for k in z.files if k.startswith("in__")}
        self.outputs = {k[5:]: z[k] for k in z.files if k.startswith("out__")}

    def state_dicts(self):
        """Rebuild the weights from (seed, tweaks) and verify the stored checksum."""
        n = self.meta.get("n_models", 1)
        if self.meta.get("weights_file"):      # trained weights (fixtures G11): stored once, shared by the fixtures
            z = np.load(os.path.join(HERE, self.meta["weights_file"]))
            sds = [{k[len(m) + 2:]: z[k].copy() for k in z.files if k.startswith(m + "__")} for m in ("coarse", "fine")[:n]]
        else:
            arch = {k: self.meta[k] for k in ("in_xyz", "in_dir") if k in self.meta}      # (G16: fewer encoding bands)
            sds = W.make_state_dict(self.meta["seed"], n, **arch)
        want = self.meta["checksum"]
        want = want if isinstance(want, list) else [want]
        for sd, c in zip(sds, want):
            W.apply_tweaks(sd, self.meta.get("tweaks", []))
            got = W.checksum(sd)
            assert abs(got - c) <= 1e-9 * max(1.0, abs(c)), \
                f"{self.name}: rebuilt weights differ from the fixture's ({got} vs {c})"
        return sds


def names(prefix=""):
    return sorted(f[:-4] for f in os.listdir(HERE) if f.endswith(".npz") and f.startswith(prefix))


# Keys whose reference value is itself noise-dominated in fp32 (normalised autograd
# gradient of the density; measured floors are stored in the fixtures, see
# DESIGN.md "tolerances").  They are compared with a floor-aware tolerance.
GRAD_NORMAL_KEYS = ("normal_coarse", "normal_fine", "surface_normal_grad_coarse",
                    "surface_normal_grad_fine", "normal_dif_coarse", "normal_dif_fine")
# Per-sample tensors evaluated at the fine sample positions: positions are not
# stable at 1e-4 (inverse-CDF bin flips), so they are only compared
# through the composited outputs.
PER_SAMPLE_FINE = ("z_vals_fine", "weights_fine", "pred_normal_fine", "normal_fine")


def tolerance(key, meta, base=1e-4):
    floor = meta.get("floor", {}).get(key, 0.0)
    tol = max(base, 4.0 * floor)
    if key.startswith("depth") or key.startswith("x_surface") or key.startswith("z_vals"):
        tol = max(tol, base * 8.0)  # depth is compared relative to far = 8
    if key in GRAD_NORMAL_KEYS:
        tol = max(tol, 2e-2)
    return tol


# ---- fixtures G17 (tests/golden/make_golden_tcnn.py): the reference's models/mirror_nerf_tcnn.py with stand-in encoders
TCNN_MLP_NAMES = ("sigma_net.0.weight", "sigma_net.1.weight", "color_net.0.weight", "color_net.1.weight", "color_net.2.weight",
                  "normal_net.0.weight", "normal_net.1.weight", "is_mirror_net.0.weight", "is_mirror_net.0.bias",
                  "is_mirror_net.2.weight", "is_mirror_net.2.bias")


def tcnn_weights(fx, prefix, which=0):
    """State dict (this repository's names, numpy) of one hash-grid model of a G17 fixture: the stored MLPs + the table
    rebuilt from its seed; `_cfg` = the level geometry (oracle.hashgrid_config)."""
    from oracle import mirror_nerf_oracle as O
    t = fx.meta["table"]
    cfg = O.hashgrid_config(t["bound"])
    seed = t["table_seeds"][which] if "table_seeds" in t else t["table_seed"]
    w = {k: fx.inputs[prefix + k].copy() for k in TCNN_MLP_NAMES}
    if t.get("keep_levels") == "smooth":
        w["encoder.embeddings"] = W.make_smooth_tcnn_table(t["bound"], seed, t["table_scale"])
    else:
        w["encoder.embeddings"] = W.make_tcnn_table(cfg["offsets"][-1], seed, t["table_scale"], t.get("keep_levels"), cfg["offsets"])
    w["_cfg"] = cfg
    return w


def table_grad_summary(g, cfg, idx):
    """Per-level [sum, norm, absmax] of a table gradient (n_entries, 2) and its values at the fixture's flat indices."""
    g = np.asarray(g, np.float64).reshape(-1, 2)
    lv = np.stack([[g[a:b].sum(), np.sqrt((g[a:b] ** 2).sum()), np.abs(g[a:b]).max()]
                   for a, b in zip(cfg["offsets"][:-1], cfg["offsets"][1:])])
    return lv, g.reshape(-1)[idx], int(np.count_nonzero(g))
