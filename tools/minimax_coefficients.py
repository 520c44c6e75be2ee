"""Near-minimax polynomial coefficients for the fp64 exp / log kernels of csrc/fm_device_math.hpp.

Chebyshev interpolation (within a small factor of the true minimax error) carried out in 60-digit decimal arithmetic,
coefficients rounded to fp64 and the error re-measured with the ROUNDED coefficients on a dense grid.

    exp:  e^r = 1 + r + r²/2 + r³·q(r),  |r| <= ln2/2 (+ margin),  q of degree 7   (replaces the degree-11 Taylor tail)
    log:  log m = 2s·(1 + z·g(z)),  s = (m-1)/(m+1), z = s² <= 0.02945,  g of degree 5 (replaces 1/3 … 1/17)

Run:  python tools/minimax_coefficients.py      (prints the C initialisers used in fm_device_math.hpp)
"""
from decimal import Decimal as D, getcontext
import math

getcontext().prec = 60
PI = D("3.14159265358979323846264338327950288419716939937510582097494")


def cos_dec(x):
    getcontext().prec += 5
    s, term, k, x2 = D(0), D(1), 0, x * x
    while abs(term) > D(10) ** -(getcontext().prec - 2):
        s += term
        k += 2
        term = -term * x2 / (k * (k - 1))
    getcontext().prec -= 5
    return +s


def solve(a, b):
    n = len(b)
    a = [row[:] + [b[i]] for i, row in enumerate(a)]
    for c in range(n):
        p = max(range(c, n), key=lambda r: abs(a[r][c]))
        a[c], a[p] = a[p], a[c]
        for r in range(c + 1, n):
            f = a[r][c] / a[c][c]
            for k in range(c, n + 1):
                a[r][k] -= f * a[c][k]
    x = [D(0)] * n
    for r in range(n - 1, -1, -1):
        x[r] = (a[r][n] - sum(a[r][k] * x[k] for k in range(r + 1, n))) / a[r][r]
    return x


def cheb_fit(fn, lo, hi, deg):
    """Monomial coefficients (in the original variable) of the interpolant at the deg+1 Chebyshev nodes of [lo, hi]."""
    n = deg + 1
    nodes = [(lo + hi) / 2 + (hi - lo) / 2 * cos_dec(PI * (2 * k + 1) / (2 * n)) for k in range(n)]
    return solve([[x ** j for j in range(n)] for x in nodes], [fn(x) for x in nodes])


def horner(c, x):
    s = D(0)
    for a in reversed(c):
        s = s * x + a
    return s


def q_exp(r):
    return (r.exp() - 1 - r - r * r / 2) / (r * r * r)


def g_log(z):
    s = z.sqrt()
    h = ((1 + s) / (1 - s)).ln() / (2 * s)
    return (h - 1) / z


def report(name, fn, lo, hi, deg, rel_scale):
    c = cheb_fit(fn, lo, hi, deg)
    cf = [float(x) for x in c]
    cd = [D(x) for x in cf]
    worst = D(0)
    m = 4001
    for i in range(m):
        x = lo + (hi - lo) * D(i) / (m - 1)
        if x == 0:
            continue
        worst = max(worst, abs(horner(cd, x) - fn(x)) * rel_scale(x))
    print(f"// {name}: degree {deg}, max relative error of the function value {float(worst):.3e} = 2^{math.log2(float(worst)):.1f}")
    for k in reversed(range(len(cf))):
        print(f"    {cf[k]!r},   // {float.hex(cf[k])}  x^{k}")
    return cf


if __name__ == "__main__":
    c = D("0.3466") + D("0.0002")                   # ln2/2 = 0.34657…, with a margin for the rounding of k
    # contribution of q to e^r (~1) is r³·q
    report("exp q(r)", q_exp, -c, c, 7, lambda r: abs(r) ** 3 / r.exp())
    Z = D("0.02945")                                # s <= (sqrt2-1)/(sqrt2+1) = 0.17157…, s² = 0.029437
    # contribution of g to log m = 2s(1 + z g) relative to log m (~2s) is z·g
    report("log g(z)", g_log, D(0), Z, 5, lambda z: z)
    report("log g(z)", g_log, D(0), Z, 6, lambda z: z)
