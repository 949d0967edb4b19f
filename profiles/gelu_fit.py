#!/usr/bin/env python3
"""Fit of the packed-fp32 GELU used by the 16-bit GEMM epilogues (genconvit_amd/csrc/gemm.h: gelu_pk).

gelu(x) = max(x, 0) - h(min(|x|, c)),  h(a) = a * 0.5 * erfc(a / sqrt 2): h is a smooth bump that decays to 0, so a
plain polynomial in t = 2a/c - 1 fits it without the |x| error amplification an erf polynomial has.  Lawson-weighted
least squares on Chebyshev nodes -> near-minimax coefficients; the script prints the monomial coefficients (highest
degree LAST) and the fp32-Horner error of the full GELU over [-8, 8].
"""
import numpy as np
from numpy.polynomial import chebyshev as Ch
from scipy.special import erf, erfc

C, DEG = 4.5, 10


def fit(c, deg, iters=400):
    t = np.unique(np.concatenate([np.cos(np.linspace(0, np.pi, 6001)), np.linspace(-1, 1, 6001)]))
    a = (t + 1) * 0.5 * c
    tgt = a * 0.5 * erfc(a / np.sqrt(2))
    A = Ch.chebvander(t, deg)
    w = np.ones_like(a)
    best = None
    for _ in range(iters):
        co, *_ = np.linalg.lstsq(A * w[:, None], tgt * w, rcond=None)
        err = np.abs(A @ co - tgt)
        if best is None or err.max() < best[0]:
            best = (err.max(), co.copy())
        w = w * (0.3 + err / err.max())
        w /= w.max()
    return best


if __name__ == "__main__":
    e, co = fit(C, DEG)
    mono = Ch.cheb2poly(co)
    print("fit max error %.3e" % e)
    print("coefficients t^0..t^%d:" % DEG, ", ".join("%.9ef" % m for m in mono))
    xs = np.linspace(-8, 8, 400001).astype(np.float32)
    a = np.minimum(np.abs(xs), np.float32(C))
    t = (a * np.float32(2 / C) - np.float32(1)).astype(np.float32)
    p = np.full_like(xs, np.float32(mono[-1]))
    for k in range(DEG - 1, -1, -1):
        p = (p * t + np.float32(mono[k])).astype(np.float32)
    out = (np.maximum(xs, 0) - p).astype(np.float32)
    true = 0.5 * xs.astype(np.float64) * (1 + erf(xs.astype(np.float64) / np.sqrt(2)))
    print("fp32 Horner |gelu error| max over [-8, 8]: %.3e" % np.abs(out - true).max())
