"""Fit and fp32 emulation of the path's erf-GELU (pmce_amd/csrc/common.hpp gelu_erf); CPU only, needs numpy + scipy + mpmath.

    gelu(x) = max(x, 0) - |x| * 2^-(|x| * R(|x|) + 1),   R(a) ~ -log2(erfc(a / sqrt 2)) / a  (degree 7, monomial basis, Horner)

The fit minimises the absolute error of the GELU (weight = d gelu / d R) by iteratively re-weighted least squares on [0, 6] with a
vanishing weight out to 20 that only keeps a * R(a) growing; the emulation runs the exact instruction sequence in fp32 (every fma
rounded once) over [-12, 12] and out to +-1e38, next to the Abramowitz-Stegun form it replaces.
"""
import numpy as np
from mpmath import mp, erfc as mperfc, log, sqrt
from scipy.special import erfc

mp.dps = 40


def r_true(a):
    return np.array([float(-log(mperfc(mp.mpf(v) / sqrt(2))) / log(2) / mp.mpf(v)) for v in a])


def fit(deg=7, A=6.0, Afar=20.0, floor=1e-13, iters=60):
    n = 3000
    a1 = (np.cos(np.pi * (np.arange(n) + 0.5) / n) + 1) * A / 2
    a = np.concatenate([a1[a1 > 1e-9], np.linspace(A, Afar, 400)[1:]])
    target = r_true(a)
    e = np.exp2(-target * a)
    wt = np.maximum(e * np.log(2) * a * np.maximum(a, 0.3) / 2, floor)
    V = np.vander(a, deg + 1, increasing=True)
    w = wt.copy()
    for _ in range(iters):  # Lawson-style re-weighting towards the minimax solution
        c, *_ = np.linalg.lstsq(V * w[:, None], target * w, rcond=None)
        err = np.abs((V @ c - target) * wt)
        w = w * (1 + 3 * err / err.max())
        w /= w.max()
    return c


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def gelu_new32(x, c32):
    x = x.astype(np.float32)
    a = np.abs(x)
    with np.errstate(all="ignore"):
        p = np.full_like(a, c32[7])
        for k in range(6, -1, -1):
            p = fma(p, a, np.full_like(a, c32[k]))
        t = fma(p, a, np.full_like(a, np.float32(1)))
        e = np.exp2(-t.astype(np.float64)).astype(np.float32)
        return fma(-a, e, np.maximum(x, np.float32(0)))


def gelu_as32(x):  # the form used until round 4
    x = x.astype(np.float32)
    one = lambda v, like: np.full_like(like, np.float32(v))
    z = (x * np.float32(0.70710678118654752440)).astype(np.float32)
    az = np.abs(z)
    t = (np.float32(1) / fma(az, one(0.3275911, az), one(1, az))).astype(np.float32)
    p = fma(t, one(1.061405429, t), one(-1.453152027, t))
    for cc in (1.421413741, -0.284496736, 0.254829592):
        p = fma(p, t, one(cc, t))
    e = np.exp2(((az * az).astype(np.float32) * np.float32(-1.44269504088896340736)).astype(np.float64)).astype(np.float32)
    r = fma(-(p * t).astype(np.float32), e, one(1, e))
    h = (x * np.float32(0.5)).astype(np.float32)
    return fma(h, np.copysign(r, z), h)


if __name__ == "__main__":
    c = fit()
    c32 = [np.float32(v) for v in c]
    print("coefficients c0..c7:", ["%.9g" % float(v) for v in c32])
    xs = np.concatenate([np.linspace(-12, 12, 6000001), np.random.default_rng(0).normal(size=1000000) * 1.5,
                         -np.logspace(0.7, 38, 20000), np.logspace(0.7, 38, 20000)])
    xf = xs.astype(np.float32).astype(np.float64)
    ref = xf * 0.5 * erfc(-xf / np.sqrt(2))
    for name, g in (("new", gelu_new32(xs, c32)), ("A-S ", gelu_as32(xs))):
        err = np.abs(g - ref)
        print(name, "max abs err %.3g at x = %.4g;  max err / max(|gelu|, 1e-3) = %.3g" % (err.max(), xs[err.argmax()], (err / np.maximum(np.abs(ref), 1e-3)).max()))
