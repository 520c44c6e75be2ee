"""Shared between tests/test_aad_cpu.py and tests/test_gpu_aad.py: expressions with closed-form derivatives."""
import math

import numpy as np


def expressions():
    """name -> (f(x, y) on RandomVariables, df/dx, df/dy in numpy float64)."""
    E = {}
    E["poly"] = (lambda x, y: x.squared().mult(y).add(x.mult(3.0)).sub(y.div(2.0)),
                 lambda x, y: 2 * x * y + 3, lambda x, y: x * x - 0.5)
    E["ratio"] = (lambda x, y: x.div(y).add(y.vid(2.0)).bus(1.0).vid(x),
                  # g = 1 - (x/y + 2/y); f = x / g
                  lambda x, y: (1 - (x + 2) / y + x / y) / (1 - (x + 2) / y) ** 2,
                  lambda x, y: -x * ((x + 2) / y ** 2) / (1 - (x + 2) / y) ** 2)
    E["transcendental"] = (lambda x, y: x.mult(y).exp().add(y.log()).add(x.sqrt()).add(y.invert()).add(x.pow(2.5)),
                           lambda x, y: y * np.exp(x * y) + 0.5 / np.sqrt(x) + 2.5 * x ** 1.5,
                           lambda x, y: x * np.exp(x * y) + 1 / y - 1 / y ** 2)
    E["trig"] = (lambda x, y: x.sin().mult(y.cos()),
                 lambda x, y: np.cos(x) * np.cos(y), lambda x, y: -np.sin(x) * np.sin(y))
    E["kinks"] = (lambda x, y: x.cap(0.9).floor(0.6).add(x.sub(y).abs()).add(x.cap(y)).add(x.floor(y)),
                  lambda x, y: ((x < 0.9) & (x >= 0.6)) * 1.0 + np.sign(x - y) + (x < y) * 1.0 + (x >= y) * 1.0,
                  lambda x, y: -np.sign(x - y) + (x >= y) * 1.0 + (x < y) * 1.0)
    E["finance_ops"] = (lambda x, y: x.accrue(y, 0.5).discount(x, 0.25).addProduct(y, 2.0).addProduct(x, y).addRatio(x, y).subRatio(y, x),
                        None, None)      # derivative by finite differences in float64
    E["choose"] = (lambda x, y: x.sub(0.75).choose(x.mult(y), y.squared()),
                   lambda x, y: np.where(x - 0.75 >= 0, y, 0.0), lambda x, y: np.where(x - 0.75 >= 0, x, 2 * y))
    E["expectation"] = (lambda x, y: x.mult(y).average().mult(x),
                        None, None)      # checked through its total: see test
    return E


def finance_ops_f64(x, y):
    a = x * (1 + y * 0.5)
    a = a / (1 + x * 0.25)
    a = a + y * 2.0
    a = a + x * y
    a = a + x / y
    a = a - y / x
    return a


def black_scholes(S0, r, sigma, T, K):
    d1 = (math.log(S0 / K) + (r + 0.5 * sigma ** 2) * T) / (sigma * math.sqrt(T))
    d2 = d1 - sigma * math.sqrt(T)
    N = lambda z: 0.5 * (1 + math.erf(z / math.sqrt(2)))
    price = S0 * N(d1) - K * math.exp(-r * T) * N(d2)
    delta = N(d1)
    vega = S0 * math.exp(-0.5 * d1 * d1) / math.sqrt(2 * math.pi) * math.sqrt(T)
    return price, delta, vega
