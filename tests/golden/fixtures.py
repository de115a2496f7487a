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
