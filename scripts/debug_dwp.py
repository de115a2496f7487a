#!/usr/bin/env python3
"""Stage-by-stage check of the operand-plane route of the weight gradients (mnrf_dwp.h) against the fp32-row route, through
the C ABI on one batch of samples:
  1. X planes of mnrf_field_forward_train(MNRF_TRAIN_PLANES), decoded on the host, vs the fp32 rows of the same forward;
  2. dY planes of mnrf_field_backward_planes (x 2^-K) vs the fp32 dY rows of mnrf_field_backward;
  3. the 32 parameter gradients of mnrf_dw_planes vs those of mnrf_field_backward, per parameter, for one and two evaluations.
Prints the largest relative error of every section / parameter: a producer bug shows in 1-2, a GEMM / finish bug only in 3."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mirror_nerf_amd as M  # noqa: E402
from mirror_nerf_amd import _lib  # noqa: E402
from mirror_nerf_amd.weights import PARAM_NAMES, PARAM_SHAPES, packed_of  # noqa: E402

dev = torch.device("cuda", 0)
L = _lib.lib()
p = _lib.ptr
f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731

SEC_X = dict(ENC=(0, 64), FIN=(2112, 256), DIRE=(2368, 32), HD=(2400, 128), HN=(2528, 128), HM=(2656, 128))
SEC_X.update({f"H{i+1}": (64 + 256 * i, 256) for i in range(8)})
SEC_Y = dict(FIN=(2048, 256), DIR=(2304, 128), NRM1=(2432, 128), MIR1=(2560, 128), RGB=(2688, 16), NRM2=(2704, 16), MIR2=(2720, 16))
SEC_Y.update({f"L{i+1}": (256 * i, 256) for i in range(8)})
SAVE_FLOATS, DY_FLOATS = 2784, 2736


def decode(planes_u8, n_fb, B):
    """[sb][fb][hi|lo][32 rows][16] f16 -> (B, 16 n_fb) float64"""
    a = planes_u8.cpu().numpy().view(np.float16)
    n_sb = a.size // (n_fb * 2 * 512)
    a = a.reshape(n_sb, n_fb, 2, 32, 16).astype(np.float64)
    v = a[:, :, 0] + a[:, :, 1]                       # (sb, fb, row, c)
    return v.transpose(0, 2, 1, 3).reshape(n_sb * 32, n_fb * 16)[:B], a


def rows(buf, off, width, B):
    return buf[off * B: off * B + width * B].view(B, width).cpu().numpy().astype(np.float64)


def rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def run(B, seed):
    torch.manual_seed(seed)
    model = M.MirrorNeRF(in_channels_xyz=63, in_channels_dir=27, predict_normal=True, predict_mirror_mask=True).to(dev)
    with torch.no_grad():
        model.sigma.weight.mul_(20.0)
    packed = packed_of(model)
    xyz = (torch.rand(B, 3, device=dev) * 6 - 3).contiguous()
    d = torch.nn.functional.normalize(torch.randn(B, 3, device=dev), dim=1)
    de = M.Embedding(4)(d)
    outs, saves = {}, {}
    for mode in ("rows", "planes"):
        sigma, rgb, pn, mir, normal = f(B), f(B, 3), f(B, 3), f(B), f(B, 3)
        if mode == "rows":
            sx = f(L.mnrf_train_save_floats(B))
        else:
            sx = torch.zeros(L.mnrf_train_planes_bytes(B), dtype=torch.uint8, device=dev)
        sm = torch.zeros(L.mnrf_train_mask_words(B), dtype=torch.int64, device=dev)
        si, sj = f(B), f(B)
        flags = _lib.MNRF_SPLIT_F16 | (_lib.MNRF_TRAIN_PLANES if mode == "planes" else 0)
        _lib.check(L.mnrf_field_forward_train(p(packed), B, p(xyz), 3, None, None, 1, p(de), 27, p(sigma), p(rgb), p(pn), p(mir),
                                              p(normal), p(sx), p(sm), p(si), p(sj), flags, _lib.stream()), "fwd " + mode)
        outs[mode] = (sigma, rgb, pn, mir, normal)
        saves[mode] = (sx, sm, si, sj)
    torch.cuda.synchronize()
    for a, b, n in zip(outs["rows"], outs["planes"], ("sigma", "rgb", "pn", "mir", "normal")):
        assert torch.equal(a, b), f"forward output {n} differs between the two save formats"
    assert torch.equal(saves["rows"][1], saves["planes"][1]), "relu masks differ"
    X, _ = decode(saves["planes"][0], 174, B)
    print(f"B={B}: X planes vs fp32 rows (largest relative error per section; f16 hi+lo carries ~2^-21):")
    for name, (off, w) in sorted(SEC_X.items()):
        r = rows(saves["rows"][0], off, w, B)
        print(f"   {name:5s} {rel(X[:, off:off + w], r):.2e}   |max| {np.abs(r).max():.3g}")
    # ---- backward
    torch.manual_seed(seed + 1)
    scale = 10.0 ** (torch.rand(B, device=dev) * 8 - 8)          # seeds over 8 orders of magnitude
    g_sigma = torch.randn(B, device=dev) * scale
    g_rgb, g_pn, g_m = torch.randn(B, 3, device=dev) * scale[:, None], torch.randn(B, 3, device=dev) * scale[:, None], torch.randn(B, device=dev) * scale
    sigma, rgb, pn, mir, normal = outs["rows"]
    grads = {}
    ws = f(L.mnrf_train_workspace_floats(B))
    d_r = [f(*PARAM_SHAPES[n]) for n in PARAM_NAMES]
    arr_r = (ctypes.c_void_p * 32)(*[t.data_ptr() for t in d_r])
    dx_r, dd_r = f(B, 3), f(B, 32)
    _lib.check(L.mnrf_field_backward(p(packed), B, p(xyz), 3, None, None, 1, p(g_sigma), p(g_rgb), p(g_pn), p(g_m), p(rgb), p(pn), p(mir),
                                     p(saves["rows"][0]), p(saves["rows"][1]), p(saves["rows"][2]), p(ws), arr_r, p(dx_r), p(dd_r), None,
                                     _lib.MNRF_SPLIT_F16, _lib.stream()), "bwd rows")
    dy = torch.zeros(L.mnrf_train_dy_planes_bytes(B), dtype=torch.uint8, device=dev)
    seedmax = torch.zeros(1, dtype=torch.int32, device=dev)
    dx_p, dd_p = f(B, 3), f(B, 32)
    _lib.check(L.mnrf_field_backward_planes(p(packed), B, p(xyz), 3, None, None, 1, p(g_sigma), p(g_rgb), p(g_pn), p(g_m), p(rgb), p(pn),
                                            p(mir), p(saves["planes"][1]), p(saves["planes"][2]), p(dy), p(seedmax), p(dx_p), p(dd_p), None,
                                            0, _lib.stream()), "bwd planes")
    torch.cuda.synchronize()
    assert torch.equal(dx_r, dx_p) and torch.equal(dd_r, dd_p), "d_xyz / d_dir differ between the two routes"
    bits = int(seedmax.item()) & 0xffffffff
    E = (bits >> 23) & 0xff
    K = (0 if E in (0, 255) else 6 - (E - 127)) + 4       # + PL_BOOST_LOG2 (mnrf_dwp.h)
    mx = np.frombuffer(np.uint32(bits).tobytes(), dtype=np.float32)[0]
    print(f"   seed maximum {mx:.4g} -> K = {K}")
    Y, _ = decode(dy, 172, B)
    Y = Y * 2.0 ** (-K)
    print("   dY planes x 2^-K vs fp32 rows (relative to the section's largest entry):")
    for name, (off, w) in sorted(SEC_Y.items()):
        r = rows(ws, off, w, B)
        print(f"   {name:5s} {rel(Y[:, off:off + w], r):.2e}   |max| {np.abs(r).max():.3g}")
    print(f"   SIG   {rel(Y[:, 2736], g_sigma.cpu().numpy().astype(np.float64)):.2e}")
    # ---- weight gradients: one evaluation, then the same evaluation twice (= 2 x)
    for n_eval in (1, 2):
        d_p = [f(*PARAM_SHAPES[n]) for n in PARAM_NAMES]
        arr_p = (ctypes.c_void_p * 32)(*[t.data_ptr() for t in d_p])
        xs = (ctypes.c_void_p * n_eval)(*[saves["planes"][0].data_ptr()] * n_eval)
        ys = (ctypes.c_void_p * n_eval)(*[dy.data_ptr()] * n_eval)
        bs = (ctypes.c_int64 * n_eval)(*[B] * n_eval)
        sms = (ctypes.c_void_p * n_eval)(*[seedmax.data_ptr()] * n_eval)
        wsp = f(L.mnrf_dw_planes_workspace_floats(n_eval, bs))
        _lib.check(L.mnrf_dw_planes(n_eval, xs, ys, bs, sms, p(wsp), arr_p, 0, _lib.stream()), "dw planes")
        torch.cuda.synchronize()
        worst = 0.0
        print(f"   weight gradients, {n_eval} evaluation(s): planes vs rows, relative to each tensor's largest entry")
        for n, a, b in zip(PARAM_NAMES, d_p, d_r):
            e = float((a - n_eval * b).abs().max() / ((n_eval * b).abs().max() + 1e-30))
            worst = max(worst, e)
            flag = "  <-- " if e > 1e-4 else ""
            print(f"      {n:32s} {e:.2e}{flag}")
        print(f"   worst {worst:.2e}")


for B, seed in ((200, 0), (1000, 1), (4096 + 77, 2)):
    run(B, seed)
