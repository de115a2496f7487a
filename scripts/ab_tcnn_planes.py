"""A/B on the GPU: hash-grid field, one 32768-ray chunk of fine-pass samples -- one-launch form vs level-major encoding planes."""
import sys
import torch
sys.path.insert(0, ".")
import mirror_nerf_amd as M
from mirror_nerf_amd import mirror_nerf as MN, synthetic as SY
dev = torch.device("cuda", 0)
torch.manual_seed(0)
models = {k: M.MirrorNeRFTcnn(encoding="hashgrid", bound=6.0, predict_normal=True, predict_mirror_mask=True).to(dev) for k in ("coarse", "fine")}
emb = {"xyz": M.Embedding(0), "dir": M.Embedding(0)}
rays = SY.device_rays(800, 800, dev)[300 * 800:300 * 800 + 32768].contiguous()
with torch.no_grad():
    rc = M.render_rays(models, emb, rays, 64, False, 0, 0, 128, 32768, test_time=True, compute_normal=False)
zf, zc = rc["z_vals_fine"].contiguous(), rc["z_vals_coarse"].contiguous()
m = models["fine"]


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


ref = None
for f16 in (False, True):
    m.mlp_f16 = f16
    for planes in (False, True):
        m.enc_planes_min = 1 if planes else 1 << 62
        with torch.no_grad():
            o = m.field(zf.numel(), rays=rays, z_vals=zf, spr=zf.shape[1])
            t = timed(lambda: m.field(zf.numel(), rays=rays, z_vals=zf, spr=zf.shape[1]))
            ts = timed(lambda: m.field(zc.numel(), rays=rays, z_vals=zc, spr=zc.shape[1], sigma_only=True))
        if ref is None:
            ref = o
        d = max(float((o[k] - ref[k]).abs().max()) for k in ("sigma", "rgb", "is_mirror"))
        print(f"f16={f16} planes={planes}: full 6.29 M samples {t:.3f} ms, sigma-only 2.1 M samples {ts:.3f} ms, max diff to the first variant {d:.2e}")
for B in (65536, 196608):
    z = zf.reshape(-1)[:B].reshape(-1, 192).contiguous() if B % 192 == 0 else zf[:B // 192 + 1].contiguous()
    r = rays[:z.shape[0]].contiguous()
    m.mlp_f16 = False
    for planes in (False, True):
        m.enc_planes_min = 1 if planes else 1 << 62
        with torch.no_grad():
            t = timed(lambda: m.field(z.numel(), rays=r, z_vals=z, spr=z.shape[1]), 20)
        print(f"B={z.numel()} planes={planes}: {t*1e3:.1f} us")
