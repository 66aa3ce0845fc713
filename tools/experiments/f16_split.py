#!/usr/bin/env python
"""Numerical experiment behind the fp16 split product (trackformer_amd/csrc/split_product.h, SP = 16; CPU, numpy only):

    python tools/experiments/f16_split.py

x . w as   xl.(wh 2^-11) + xh.wl + xh.wh   with   xh = f16(x / 16), xl = f16((x / 16 - xh) 2^11), wh = f16(w t_n), wl = f16(w t_n - wh)
(t_n: the power of two that puts the largest |w| of output channel n into [2^13, 2^14)), result times 16 / t_n -- against float64,
beside numpy's sgemm, the six-term and the three-term bf16 products and the same fp16 product WITHOUT the scaling (plain hi / lo
pieces: what goes wrong when activations are small).  The partial products are summed in float64 here, so the columns show the error
of the REPRESENTATION; on the matrix cores the sum is rounded to fp32 per MFMA like sgemm's (its column is the scale to read the
others against).  Error measure: max and rms over the outputs of |y - exact| / sum_k |x_k||w_k|."""
import numpy as np

rng = np.random.default_rng(0)


def bf16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)


def split_bf(x, n):
    out, r = [], x.astype(np.float32)
    for _ in range(n):
        p = bf16(r)
        out.append(p)
        r = (r - p).astype(np.float32)
    return out


def f16_activation(x):
    xs = (x * np.float32(0.0625)).astype(np.float32)
    hi = xs.astype(np.float16)
    lo = ((xs - hi.astype(np.float32)) * np.float32(2048)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


def f16_weight(w):   # w [K, N]: channels are columns here
    amax = np.abs(w).max(0)
    _, e = np.frexp(amax)
    t = np.ldexp(np.float32(1), 14 - e).astype(np.float32)
    ws = (w * t).astype(np.float32)
    hi = ws.astype(np.float16)
    lo = (ws - hi.astype(np.float32)).astype(np.float16)
    hs = (hi.astype(np.float32) * np.float32(1 / 2048)).astype(np.float16)   # exact unless subnormal (then rounded like the kernel's)
    return hi.astype(np.float64), lo.astype(np.float64), hs.astype(np.float64), 16.0 / t.astype(np.float64)


def plain_f16(x):
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


def run(M, K, N, xs, ws, label):
    x = (rng.standard_normal((M, K)) * xs).astype(np.float32)
    w = (rng.standard_normal((K, N)) * ws).astype(np.float32)
    ref = x.astype(np.float64) @ w.astype(np.float64)
    den = np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64)

    def err(y):
        e = np.abs(y - ref) / den
        return "%.1e / %.1e" % (e.max(), np.sqrt((e ** 2).mean()))
    f = lambda a, b: a.astype(np.float64) @ b.astype(np.float64)
    xb, wb = split_bf(x, 3), split_bf(w, 3)
    six = f(xb[0], wb[0]) + f(xb[0], wb[1]) + f(xb[1], wb[0]) + f(xb[1], wb[1]) + f(xb[0], wb[2]) + f(xb[2], wb[0])
    three = f(xb[0], wb[0]) + f(xb[0], wb[1]) + f(xb[1], wb[0])
    xh, xl = f16_activation(x)
    wh, wl, whs, r = f16_weight(w)
    half = (xl @ whs + xh @ wl + xh @ wh) * r
    ph, pl = plain_f16(x)
    qh, ql = plain_f16(w)
    plain = ph @ qh + ph @ ql + pl @ qh
    print("%-34s sgemm %s | bf16 x 6 %s | bf16 x 3 %s | fp16 (scheme) %s | fp16 plain pieces %s" % (
        label, err((x @ w).astype(np.float64)), err(six), err(three), err(half), err(plain)))


if __name__ == "__main__":
    print("max / rms of |y - exact| / sum |x||w|")
    run(512, 256, 256, 1.0, 0.05, "x ~ 1, w ~ 0.05, K = 256")
    run(512, 1024, 256, 1.0, 0.02, "K = 1024")
    run(512, 2304, 256, 1.0, 0.02, "K = 2304 (3 x 3 x 256)")
    run(512, 256, 256, 0.01, 0.05, "x ~ 0.01")
    run(512, 256, 256, 1e-4, 1e-3, "x ~ 1e-4, w ~ 1e-3")
    run(512, 256, 256, 1e4, 0.05, "x ~ 1e4")
