#!/usr/bin/env python3
"""Fit of the packed-fp32 GELU used by the 16-bit GEMM epilogues (genconvit_amd/csrc/gemm.h: gelu_pk).

gelu(x) = max(x, 0) - h(min(|x|, c)),  h(a) = a * 0.5 * erfc(a / sqrt 2): h is a smooth bump that decays to 0, so a
plain polynomial in t = 2a/c - 1 fits it without the |x| error amplification an erf polynomial has.  Lawson-weighted
least squares on Chebyshev nodes -> near-minimax coefficients; the script prints the monomial coefficients (highest
degree LAST) and the fp32-Horner error of the full GELU over [-8, 8].

`gelu_fit.py --h16`: numpy simulation of GeluH16 (gemm.h) — the polynomial on the packed-fp16 pipe — for several
(degree, clamp) pairs, with the last step in fp32 (v_fma_mix, bf16 storage) or packed fp16 (fp16 storage): rms / max error
of the STORED fp16 activation against the exact GELU, for N(0, 1.5) inputs and a sweep of [-6, 6], next to the error of
the exact GELU merely rounded to fp16.  Round 3 picked degree 8 on [0, 4] from this table.
"""
import sys
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


def h16_table():
    from scipy.special import erf as _erf
    f16 = np.float16
    fma16 = lambda a, b, c: (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f16)   # one rounding

    def sim(x32, mono, clamp, packed_finish):
        xh = x32.astype(f16)
        a = np.minimum(np.abs(xh), f16(clamp))
        t = fma16(a, np.full_like(a, f16(2 / clamp)), np.full_like(a, f16(-1)))
        co = [f16(-0.5 * m) for m in mono]                 # the polynomial carries -h/2 (gemm.h)
        p = fma16(np.full_like(t, co[-1]), t, np.full_like(t, co[-2]))
        for k in range(len(mono) - 3, -1, -1):
            p = fma16(p, t, np.full_like(t, co[k]))
        if packed_finish:                                  # v_pk_max_i16 + v_pk_fma_f16: x rounded to fp16 first
            return fma16(p, np.full_like(p, f16(2)), np.maximum(xh, f16(0)))
        return (np.maximum(x32, 0).astype(np.float64) + 2 * p.astype(np.float64)).astype(np.float32).astype(f16)

    truth = lambda x32: 0.5 * x32.astype(np.float64) * (1 + _erf(x32.astype(np.float64) / np.sqrt(2)))
    rng = np.random.default_rng(0)
    xs = [("N(0,1.5)", (rng.standard_normal(4_000_000) * 1.5).astype(np.float32)),
          ("sweep[-6,6]", np.linspace(-6, 6, 2_000_001).astype(np.float32))]
    for name, x in xs:
        tr = truth(x)
        ex = np.abs(tr.astype(np.float32).astype(f16).astype(np.float64) - tr)
        print("exact GELU rounded to fp16, %-12s rms %.3e max %.3e" % (name, np.sqrt((ex ** 2).mean()), ex.max()))
    for deg, clamp in [(10, 4.5), (9, 4.5), (8, 4.5), (8, 4.25), (8, 4.0), (7, 4.0)]:
        e, co = fit(clamp, deg, iters=200)
        mono = Ch.cheb2poly(co)
        for pf in (False, True):
            out = []
            for name, x in xs:
                err = sim(x, mono, clamp, pf).astype(np.float64) - truth(x)
                out.append("%s rms %.3e max %.3e" % (name, np.sqrt((err ** 2).mean()), np.abs(err).max()))
            print("degree %2d on [0, %.2f] (fit %.2e) %s | %s" % (deg, clamp, e, "packed fp16 finish" if pf else "fp32 finish       ", " | ".join(out)))


if __name__ == "__main__":
    if "--h16" in sys.argv:
        h16_table()
        sys.exit(0)
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
