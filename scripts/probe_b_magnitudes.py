"""How large are the density-gradient signals b_i = d sigma / d h_i (pre-mask) on trained weights?  They bound the boost the
second-order planes can give them (mnrf_dwp.h).  CPU, float64."""
import os
import sys

import numpy as np
import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(root, "tests"))
import torch_ref as TR  # noqa: E402

fx = np.load(os.path.join(root, "tests", "golden", "g11_trained_weights.npz"))
for which in ("coarse", "fine"):
    w = {k[len(which) + 2:]: torch.from_numpy(fx[k]).double() for k in fx.files if k.startswith(which + "__")}
    if not w:
        print("keys", fx.files[:6]); break
    torch.manual_seed(0)
    x = (torch.rand(4096, 3, dtype=torch.float64) * 2 - 1) * 1.5
    enc = TR.embed(x, 10)
    hs, h = [], enc
    for i in range(8):
        if i == 4:
            h = torch.cat([enc, h], -1)
        h = torch.relu(h @ w[f"xyz_encoding_{i+1}.0.weight"].T + w[f"xyz_encoding_{i+1}.0.bias"])
        h.retain_grad() if h.requires_grad else None
        hs.append(h)
    # b_8 = w_sigma * mask_8 ; b_i = (W_{i+1}^T b_{i+1}) * mask_i
    b = w["sigma.weight"][0][None, :] * (hs[7] > 0)
    out = [b]
    for i in range(7, 0, -1):
        W = w[f"xyz_encoding_{i+1}.0.weight"]
        if i == 4:
            W = W[:, 63:]
        b = (b @ W) * (hs[i - 1] > 0)
        out.append(b)
    out = out[::-1]
    print(which, "max |b_i|, i = 1..8:", " ".join(f"{float(t.abs().max()):.3g}" for t in out),
          "| median of nonzero:", " ".join(f"{float(t[t != 0].abs().median()):.2g}" for t in out))
