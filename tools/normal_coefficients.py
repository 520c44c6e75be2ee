"""Polynomial coefficients of the Box–Muller transform of the normal-increment generator (oracle/philox_normal.c,
csrc/kernels.hip: fm_bm_kernel), and their error measured in SIMULATED fp32 arithmetic (numpy float32, fused multiply-adds
emulated in float64 — exact for one fp32 FMA: 24 x 24 bit product + 24 bit addend fits 53 bits only approximately, so the
emulation rounds twice in rare cases; the measured error bounds are unaffected at the 1e-8 level).

  radius² = -2 ln u,  u = m · 2^-(lz+1),  m in [1, 2]:   radius² = lz · 2ln2 + Q(t),  t = m - 1.5,  Q(t) = 2ln2 - 2 ln(1.5 + t)
  (cos x, sin x) on x in [0, π/4]:  sin x = x + x·x²·S(x²),  cos x = 1 + x²·C(x²)

Chebyshev interpolation in 60-digit arithmetic (tools/minimax_coefficients.py: cheb_fit), coefficients rounded to fp32.

    python tools/normal_coefficients.py          prints the C initialisers and the measured errors
"""
import math
import sys
import os
from decimal import Decimal as D, getcontext

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from minimax_coefficients import cheb_fit, cos_dec, PI  # noqa: E402

getcontext().prec = 60
LN2 = D(2).ln()


def sin_dec(x):
    return cos_dec(PI / 2 - x)


def q_radius(t):
    return 2 * LN2 - 2 * (D("1.5") + t).ln()


def s_sin(z):                      # (sin x - x) / x³ as a function of z = x²
    if z == 0:
        return -D(1) / 6
    x = z.sqrt()
    return (sin_dec(x) - x) / (x * z)


def c_cos(z):                      # (cos x - 1) / x² as a function of z = x²
    if z == 0:
        return -D(1) / 2
    x = z.sqrt()
    return (cos_dec(x) - 1) / z


def f32(c):
    return [float(np.float32(float(x))) for x in c]


def horner32(c, x):
    """Horner in fp32 with fused multiply-adds (highest coefficient first)."""
    acc = np.full_like(x, np.float32(c[0]), dtype=np.float32)
    for a in c[1:]:
        acc = (acc.astype(np.float64) * x.astype(np.float64) + np.float64(np.float32(a))).astype(np.float32)
    return acc


def main():
    # ---- radius²: degree chosen so that the approximation error sits at the fp32 rounding level
    for deg in (8, 9):
        c = f32(cheb_fit(q_radius, D("-0.5"), D("0.5"), deg))
        t = np.linspace(-0.5, 0.5, 2_000_001).astype(np.float32)
        got = horner32(list(reversed(c)), t).astype(np.float64)
        want = 2 * math.log(2) - 2 * np.log(1.5 + t.astype(np.float64))
        print(f"// Q(t) = 2ln2 - 2ln(1.5+t), |t| <= 0.5, degree {deg}: max abs error in fp32 Horner {np.abs(got - want).max():.3e}, Q(0.5) = {got[-1]!r}")
        for k in reversed(range(len(c))):
            print(f"    {np.float32(c[k])!r:>16}f,   // {float.hex(c[k])}  t^{k}")
    # ---- sin / cos on [0, π/4]
    zmax = (PI / 4) ** 2 * (1 + D(2) ** -20)
    cs = f32(cheb_fit(s_sin, D(0), zmax, 2))        # sin: degree 7 in x
    cc = f32(cheb_fit(c_cos, D(0), zmax, 3))        # cos: degree 8 in x
    x = np.linspace(0, math.pi / 4, 2_000_001).astype(np.float32)
    x2 = (x.astype(np.float64) * x.astype(np.float64)).astype(np.float32)
    sp = horner32(list(reversed(cs)), x2)
    sp = (sp.astype(np.float64) * x2.astype(np.float64)).astype(np.float32)                         # sp * x2
    sinx = (x.astype(np.float64) * sp.astype(np.float64) + x.astype(np.float64)).astype(np.float32)  # fma(x, sp, x)
    cp = horner32(list(reversed(cc)), x2)
    cosx = (cp.astype(np.float64) * x2.astype(np.float64) + 1.0).astype(np.float32)                 # fma(cp, x2, 1)
    print(f"// sin on [0, π/4], x + x·x²·S(x²), S of degree 2: max abs error {np.abs(sinx - np.sin(x.astype(np.float64))).max():.3e}")
    for k in reversed(range(len(cs))):
        print(f"    {np.float32(cs[k])!r:>16}f,   // {float.hex(cs[k])}  z^{k}")
    print(f"// cos on [0, π/4], 1 + x²·C(x²), C of degree 3: max abs error {np.abs(cosx - np.cos(x.astype(np.float64))).max():.3e}")
    for k in reversed(range(len(cc))):
        print(f"    {np.float32(cc[k])!r:>16}f,   // {float.hex(cc[k])}  z^{k}")
    print(f"// sin² + cos² - 1: max {np.abs(sinx.astype(np.float64) ** 2 + cosx.astype(np.float64) ** 2 - 1).max():.3e}")


if __name__ == "__main__":
    main()
