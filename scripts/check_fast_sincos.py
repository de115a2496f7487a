#!/usr/bin/env python3
"""Host-side emulation (numpy, fused multiply-adds emulated through float64) of fast_sincos() in
csrc/mnrf_field_split.inc: max |error| against double over the argument range of the positional encoding
(2^f * x, f = 0..9, |x| <= 64: ENC_RANGE_LIMIT of the kernels).  Prints ~1.1e-7."""
import numpy as np


def f32(x):
    return np.float32(x)


def fma(a, b, c):
    return np.float32(np.float64(a) * np.float64(b) + np.float64(c))


def fast(a):
    k = np.rint(np.float32(a * f32(0.6366197723675814))).astype(np.float32)
    r = fma(-k, f32(1.5703125), a)
    r = fma(-k, f32(4.8351287841796875e-4), r)
    r = fma(-k, f32(3.1385570764541626e-7), r)
    r = fma(-k, f32(6.07710062827671e-11), r)
    r2 = np.float32(r * r)
    s = fma(fma(fma(f32(-1.9515295891e-4), r2, f32(8.3321608736e-3)), r2, f32(-1.6666654611e-1)), np.float32(r2 * r), r)
    c = fma(fma(fma(f32(2.443315711809948e-5), r2, f32(-1.388731625493765e-3)), r2, f32(4.166664568298827e-2)),
            np.float32(r2 * r2), fma(f32(-0.5), r2, f32(1.0)))
    q = k.astype(np.int64) & 3
    ss = np.where(q & 1, c, s)
    cc = np.where(q & 1, s, c)
    return np.where(q & 2, -ss, ss).astype(np.float32), np.where((q + 1) & 2, -cc, cc).astype(np.float32)


x = np.random.RandomState(0).uniform(-64, 64, 2000000).astype(np.float32)
worst = 0.0
for f in range(10):
    a = np.ldexp(x, f).astype(np.float32)
    s, c = fast(a)
    worst = max(worst, np.abs(s.astype(np.float64) - np.sin(a.astype(np.float64))).max(),
                np.abs(c.astype(np.float64) - np.cos(a.astype(np.float64))).max())
print(f"fast_sincos: max |error| vs double over |a| <= {64 * 2 ** 9}: {worst:.3e}")
assert worst < 1.5e-7
