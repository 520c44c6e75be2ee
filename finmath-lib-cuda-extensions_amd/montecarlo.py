"""Small Monte-Carlo drivers written ONLY against the RandomVariable / BrownianMotion interfaces — the callers on
the other side of the hot path.  They accept any factory / Brownian motion that
implements the interface (RandomVariableHip…, or the CPU twin in the tests), exactly as finmath-lib's models accept
a RandomVariableFactory by injection (LIBORMarketModelCalibrationATMTest.java:351-358).

    black_scholes_call_mc   MonteCarloBlackScholesModelTest.java:62-85,125-157 (Euler scheme on the log state,
                            numeraire exp(r t), European call)
    heston_call_mc          BASELINE.json configs[2]: Euler full-truncation Heston driven by BrownianMotionHip
"""
from __future__ import annotations

import math


def black_scholes_call_mc(brownian_motion, initial_value, risk_free_rate, volatility, maturity, strike):
    """Value of a European call under Black–Scholes by Monte-Carlo: log-Euler scheme
    X_{i+1} = X_i + (r - σ²/2) Δt_i + σ ΔW_i, S = exp(X); payoff max(S_T - K, 0) / exp(r T).
    `maturity` must be a point of the Brownian motion's time discretisation."""
    td = brownian_motion.getTimeDiscretization()
    x = brownian_motion.getRandomVariableForConstant(math.log(initial_value))
    t, i = td.getTime(0), 0
    while t < maturity - 1e-12:
        dt = td.getTimeStep(i)
        dw = brownian_motion.getBrownianIncrement(i, 0)
        x = x.add((risk_free_rate - 0.5 * volatility * volatility) * dt).addProduct(dw, volatility)
        i += 1
        t = td.getTime(i)
    payoff = x.exp().sub(strike).floor(0.0)
    value = payoff.div(math.exp(risk_free_rate * maturity))
    return value.getAverage(), value


def black_scholes_call_analytic(initial_value, risk_free_rate, volatility, maturity, strike):
    """net.finmath.functions.AnalyticFormulas.blackScholesOptionValue (closed form)."""
    from math import erf, exp, log, sqrt
    d1 = (log(initial_value / strike) + (risk_free_rate + 0.5 * volatility ** 2) * maturity) / (volatility * sqrt(maturity))
    d2 = d1 - volatility * sqrt(maturity)
    cdf = lambda z: 0.5 * (1.0 + erf(z / sqrt(2.0)))
    return initial_value * cdf(d1) - strike * exp(-risk_free_rate * maturity) * cdf(d2)


def heston_call_mc(brownian_motion, initial_value, risk_free_rate, v0, kappa, theta, xi, rho, maturity, strike):
    """Euler full-truncation Heston:  v⁺ = max(v, 0);
        X_{i+1} = X_i + (r - v⁺/2) Δt + sqrt(v⁺) ΔW¹
        v_{i+1} = v_i + κ(θ - v⁺) Δt + ξ sqrt(v⁺) (ρ ΔW¹ + sqrt(1-ρ²) ΔW²)
    Uses factors 0 and 1 of the Brownian motion.  ξ = 0, v0 = θ reduces to Black–Scholes with σ² = θ."""
    td = brownian_motion.getTimeDiscretization()
    x = brownian_motion.getRandomVariableForConstant(math.log(initial_value))
    v = brownian_motion.getRandomVariableForConstant(v0)
    rho_c = math.sqrt(1.0 - rho * rho)
    t, i = td.getTime(0), 0
    while t < maturity - 1e-12:
        dt = td.getTimeStep(i)
        dw1 = brownian_motion.getBrownianIncrement(i, 0)
        dw2 = brownian_motion.getBrownianIncrement(i, 1)
        vp = v.floor(0.0)
        sq = vp.sqrt()
        x = x.add(risk_free_rate * dt).addProduct(vp, -0.5 * dt).addProduct(sq, dw1)
        if xi != 0.0:
            dz = dw1.mult(rho).addProduct(dw2, rho_c)
            v = v.addProduct(vp.bus(theta), kappa * dt).addProduct(sq.mult(xi), dz)
        else:
            v = v.addProduct(vp.bus(theta), kappa * dt)
        i += 1
        t = td.getTime(i)
    payoff = x.exp().sub(strike).floor(0.0)
    value = payoff.div(math.exp(risk_free_rate * maturity))
    return value.getAverage(), value
