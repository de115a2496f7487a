#!/usr/bin/env python3
"""How far is dwp_gemm_kernel from what the memory system delivers to ANY streaming reader of the same bytes?  Times (HIP events)
(a) torch's sum over a float32 view of an X-plane buffer of the training step's size, (b) a device-to-device copy of it (read +
write), (c) mnrf_dw_planes over the same sample count (two evaluations: 196 608 + 49 152 samples, the training step's shape)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mirror_nerf_amd import _lib  # noqa: E402
from mirror_nerf_amd.weights import PARAM_NAMES, PARAM_SHAPES  # noqa: E402

dev = torch.device("cuda", 0)
L, p = _lib.lib(), _lib.ptr


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


Bs = [196608, 49152]
xs = [torch.randint(0, 255, (L.mnrf_train_planes_bytes(B),), dtype=torch.uint8, device=dev) for B in Bs]
ys = [torch.randint(0, 255, (L.mnrf_train_dy_planes_bytes(B),), dtype=torch.uint8, device=dev) for B in Bs]
total = sum(t.numel() for t in xs + ys)
big = xs[0].view(torch.float32)
t = timed(lambda: big.sum())
print(f"torch sum over {big.numel() * 4 / 1e9:.2f} GB: {big.numel() * 4 / t / 1e12:.2f} TB/s")
dst = torch.empty_like(big)
t = timed(lambda: dst.copy_(big))
print(f"torch copy of the same (read + write): {2 * big.numel() * 4 / t / 1e12:.2f} TB/s")
# (d) the GEMM's own instruction: global_load_lds_dwordx4 streaming reads of the same buffer, nothing computed
for aux in (0, 2):
    for depth in (8, 16):
        t = timed(lambda: _lib.check(L.mnrf_bench_stream(ctypes.c_void_p(xs[0].data_ptr()), xs[0].numel(), aux, depth, _lib.stream()), "stream"))
        print(f"LDS-DMA streaming read ({'nt' if aux else 'default'} policy, {depth} x 1 KiB in flight per wave) of {xs[0].numel() / 1e9:.2f} GB: "
              f"{xs[0].numel() // 256 // 8192 * 8192 * 256 / t / 1e12:.2f} TB/s")
# the same instruction with the GEMM's address pattern (job 1: 32 KiB of dY + 32 KiB of X per 32-sample stage, strides of the plane
# layouts), without and with a workgroup barrier per stage; then with the chunks contiguous (stride = chunk)
nsb = xs[0].numel() // (174 * 2048)
for name, sa, sx, win in (("10 jobs over the plane strides (344 / 348 KiB)", 172 * 2048, 174 * 2048, 10),
                          ("one job over contiguous chunks", 32768, 32768, 1)):
    for barrier in (0, 1, 2, 3, 7, 11):      # bit 0: barrier per stage, bit 1: the ring's half-tile lane pattern, bits 2 / 3: bulk waits
        ns = nsb if win > 1 else min(ys[0].numel(), xs[0].numel()) // 32768
        t = timed(lambda: _lib.check(L.mnrf_bench_stream2(ctypes.c_void_p(ys[0].data_ptr()), ctypes.c_void_p(xs[0].data_ptr()), ns, sa, sx,
                                                          32768, 32768, win, barrier, _lib.stream()), "stream2"))
        print(f"LDS-DMA read (nt), 2 x 32 KiB per stage, {name}, {'barrier per stage' if barrier & 1 else 'no barrier'}{', half-tile pattern' if barrier & 2 else ''}{', stage-wise wait for all but 4 requests per wave' if barrier & 4 else ''}{', stage-wise wait for all but 12' if barrier & 8 else ''}: "
              f"{ns * win * 65536 / t / 1e12:.2f} TB/s")
n = len(Bs)
bs = (ctypes.c_int64 * n)(*Bs)
seed = [torch.zeros(1, dtype=torch.int32, device=dev) for _ in Bs]
ws = torch.empty(L.mnrf_dw_planes_workspace_floats(n, bs), dtype=torch.float32, device=dev)
d_p = [torch.empty(*PARAM_SHAPES[k], dtype=torch.float32, device=dev) for k in PARAM_NAMES]
arr = (ctypes.c_void_p * 32)(*[q.data_ptr() for q in d_p])
X = (ctypes.c_void_p * n)(*[q.data_ptr() for q in xs])
Y = (ctypes.c_void_p * n)(*[q.data_ptr() for q in ys])
S = (ctypes.c_void_p * n)(*[q.data_ptr() for q in seed])
t = timed(lambda: _lib.check(L.mnrf_dw_planes(n, X, Y, bs, S, p(ws), arr, 0, _lib.stream()), "dw"))
sb = sum((B + 127) // 128 * 4 for B in Bs)
print(f"mnrf_dw_planes: {t * 1e3:.3f} ms; unique plane bytes {total / 1e9:.2f} GB = {total / t / 1e12:.2f} TB/s; "
      f"bytes the jobs read {sb * 812 * 1024 / 1e9:.2f} GB = {sb * 812 * 1024 / t / 1e12:.2f} TB/s")

# the same launch over second-order planes only (kind 1: the trunk jobs and sigma, 562 KiB per sample block, 91 % of it in the
# 64-KiB-per-stage jobs): how much of the distance to the probe belongs to the small jobs of the first-order table
x2 = torch.randint(0, 255, (L.mnrf_train_planes2_bytes(Bs[0]),), dtype=torch.uint8, device=dev)
y2 = torch.randint(0, 255, (L.mnrf_train_dy_planes2_bytes(Bs[0]),), dtype=torch.uint8, device=dev)
bs1 = (ctypes.c_int64 * 1)(Bs[0])
kd1 = (ctypes.c_int * 1)(1)
ws1 = torch.empty(L.mnrf_dw_planes2_workspace_floats(1, bs1, kd1), dtype=torch.float32, device=dev)
t = timed(lambda: _lib.check(L.mnrf_dw_planes2(1, (ctypes.c_void_p * 1)(x2.data_ptr()), (ctypes.c_void_p * 1)(y2.data_ptr()), bs1,
                                               (ctypes.c_void_p * 1)(seed[0].data_ptr()), kd1, p(ws1), arr, 0, _lib.stream()), "dw2"))
sb1 = (Bs[0] + 127) // 128 * 4
print(f"mnrf_dw_planes2, second-order planes of {Bs[0]} samples: {t * 1e3:.3f} ms; bytes the jobs read {sb1 * 562 * 1024 / 1e9:.2f} GB = "
      f"{sb1 * 562 * 1024 / t / 1e12:.2f} TB/s")
