#!/usr/bin/env python3
"""Fixture G12: the pin-hole ray generator captured from the reference (datasets/ray_utils.py:6-53 +
datasets/blender.py:159-168: rays = [o, d, near, far]).  Build-container only:
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_rays.py
kornia's create_meshgrid is absent here; `_ref_import._create_meshgrid` restates its published definition (exact integers)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.dont_write_bytecode = True

import _ref_import as R  # noqa: E402

RU = R.ray_utils()
import torch  # noqa: E402

from make_golden import save  # noqa: E402
from oracle import mirror_nerf_oracle as O  # noqa: E402


def main():
    for name, H, W, angle, eye in (("g12_rays_37x53", 37, 53, 0.6911112, (1.3, -3.1, 2.2)),
                                   ("g12_rays_64x64", 64, 64, 0.9, (0.0, -4.0, 1.5))):
        focal = 0.5 * W / np.tan(0.5 * angle)
        pose = O.look_at_pose(eye=eye)
        dirs = RU.get_ray_directions(H, W, focal)
        o, d = RU.get_rays(dirs, torch.from_numpy(pose))
        near, far = 0.05, 8.0
        rays = torch.cat([o, d, near * torch.ones_like(o[:, :1]), far * torch.ones_like(o[:, :1])], 1)      # blender.py:163-167
        oo, od = O.get_rays(O.get_ray_directions(H, W, focal), pose)
        err = max(np.abs(oo - o.numpy()).max(), np.abs(od - d.numpy()).max())
        print(f"  {name}: oracle vs reference max-abs {err:.2e}")
        assert err <= 1e-6
        save(name, dict(H=H, W=W, focal=float(focal), near=near, far=far), {"pose": pose}, {"rays": rays.numpy()})


if __name__ == "__main__":
    main()
