"""The smile-calibration driver (host/lmm_smile.hpp; LIBORMarketModelCalibrationTest.java) on the CPU twin: the host-side
pieces — curve and products, factor reduction, implied-volatility inversion — and a short calibration against the
reference test's acceptance threshold |mean deviation| < 1e-2 (:358).  No GPU."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMILE_CPU = os.path.join(ROOT, "oracle", "host", "lmm_smile_cpu")


def run(*args):
    if not os.path.exists(SMILE_CPU):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    out = subprocess.run([SMILE_CPU, *map(str, args)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.fixture(scope="module")
def selftest():
    return run("--mode", "selftest")


def test_products_and_curve(selftest):
    # 9 smile strikes on the 5y x 10y swap + 10 ATM swaptions (:224-246); the four with exercise >= 15y reach beyond the 20y
    # LIBOR horizon and drop out (the reference swallows their exception)
    assert selftest["valid"] == [1] * 15 + [0] * 4
    par = np.array(selftest["par_swaprate"])
    assert np.all(par[:9] == par[0]) and par[12] == par[0]                      # the 5y x 10y swap appears ten times
    # independent restatement: discount factors Π 1/(1 + L δ) from the forward curve of :205-207 (first 32 values suffice)
    fwd = np.array([0.61, 0.61, 0.67, 0.73, 0.80, 0.92, 1.11, 1.36, 1.60, 1.82, 2.02, 2.17, 2.27, 2.36, 2.46, 2.52, 2.54, 2.57, 2.68, 2.82, 2.92, 2.98,
                    3.00, 2.99, 2.95, 2.89, 2.82, 2.74, 2.66, 2.59, 2.52, 2.47]) / 100
    df = np.concatenate([[1.0], np.cumprod(1.0 / (1.0 + 0.5 * fwd))])
    j = np.arange(10, 30)                                                       # periods 5.0 … 15.0
    annuity = np.sum(0.5 * df[j + 1])
    assert abs(annuity - selftest["annuity"][0]) < 1e-13
    assert abs(np.sum(fwd[j] * 0.5 * df[j + 1]) / annuity - par[0]) < 1e-15
    assert abs((df[10] - df[30]) / annuity - par[0]) < 1e-14                    # telescoping: (DF(5) − DF(15)) / annuity


def test_factor_reduction(selftest):
    f = np.array(selftest["factors"]).reshape(40, 5)
    assert np.allclose(np.sum(f * f, axis=1), 1.0, atol=1e-14)                  # rows renormalised: unit diagonal
    t = 0.5 * np.arange(40)
    full = np.exp(-0.10 * np.abs(t[:, None] - t[None, :]))
    w, v = np.linalg.eigh(full)
    top = v[:, ::-1][:, :5] * np.sqrt(w[::-1][:5])
    top /= np.linalg.norm(top, axis=1, keepdims=True)
    assert np.allclose(f @ f.T, top @ top.T, atol=1e-10)                        # same reduced correlation as LAPACK's eigenpairs
    assert np.max(np.abs(f @ f.T - full)) < 0.2                                 # 5 of 40 factors: close to the full matrix
    assert np.all(f[0] > 0)                                                     # sign convention
    # a finite-difference bump of the decay parameter must not flip a factor (every path would jump)
    g = np.array(selftest["factors_bumped"]).reshape(40, 5)
    assert np.max(np.abs(f - g)) < 1e-3                                          # continuous in the parameter (a flip would be O(1))


def test_implied_volatility_round_trip(selftest):
    rt = np.array(selftest["implied_round_trip"])
    assert rt.shape == (19 * 4, 2)
    assert np.max(np.abs(rt[:, 0] - rt[:, 1])) < 1e-9


def test_evaluation_smile_shape_and_short_calibration():
    ev = run("--paths", 2048, "--mode", "evaluate")
    vols = ev["model_volatility"]
    assert vols[15:] == [None] * 4 and all(0.2 < v < 0.6 for v in vols[:15])
    assert vols[4] == vols[12]                                                  # the ATM 5y x 10y swaption is product 4 and product 12
    assert all(a > b for a, b in zip(vols[:8], vols[1:9]))                      # displaced dynamics: implied volatility falls with the strike
    cal = run("--paths", 2048, "--max-iterations", 6)
    assert cal["products_valued"] == 15 and cal["parameters_calibrated"] == 8
    assert cal["iterations"] == 7 and cal["evaluations"] <= 7 + 8 * 7 + 1      # one trial point per iteration, a Jacobian per accepted point, the final valuation
    assert cal["rms_deviation"] < 0.5 * cal["initial_rms"]
    assert abs(cal["mean_deviation"]) < 1e-2                                    # LIBORMarketModelCalibrationTest.java:358
