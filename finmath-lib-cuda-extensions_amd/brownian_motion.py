"""BrownianMotionHip — mirror of BrownianMotionCudaWithRandomVariableCuda
(src/main/java/net/finmath/cuda/montecarlo/alternative/BrownianMotionCudaWithRandomVariableCuda.java, ``:line``).

Same constructor arguments and method set; increments are device-resident RandomVariableHip objects with
filtration time t_{i+1} (:169,:176), generated eagerly on first access (:123-130) by ONE kernel launch for all
(step, factor) vectors instead of one cuRAND call per vector (:168-178).

Additions for path sharding over GPUs: ``path_offset`` = global index of this process's first
path; the generator is counter-based, so the union of the shards equals the single-GPU stream bit for bit.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

from . import _native as N
from .random_variable import DeviceVector, RandomVariableHip, RandomVariableHipFactory


class TimeDiscretization:
    """Minimal stand-in for net.finmath.time.TimeDiscretizationFromArray (finmath-lib, not vendored):
    ``TimeDiscretization(initial, number_of_time_steps, delta_t)`` or ``TimeDiscretization(times)``."""

    def __init__(self, *args):
        if len(args) == 3:
            t0, n, dt = args
            self.times = np.array([t0 + i * dt for i in range(int(n) + 1)], dtype=np.float64)
        else:
            self.times = np.asarray(args[0], dtype=np.float64)

    def getNumberOfTimeSteps(self): return self.times.size - 1
    def getNumberOfTimes(self): return self.times.size
    def getTime(self, i): return float(self.times[i])
    def getTimeStep(self, i): return float(self.times[i + 1] - self.times[i])
    def getAsDoubleArray(self): return self.times.copy()
    def __eq__(self, o): return isinstance(o, TimeDiscretization) and np.array_equal(self.times, o.times)
    def __hash__(self): return hash(self.times.tobytes())


class BrownianMotionHip:
    def __init__(self, time_discretization, number_of_factors, number_of_paths, seed,
                 random_variable_factory=None, path_offset=0):
        self.timeDiscretization = time_discretization
        self.numberOfFactors = int(number_of_factors)
        self.numberOfPaths = int(number_of_paths)
        self.seed = int(seed)
        self.pathOffset = int(path_offset)
        self.randomVariableFactory = random_variable_factory or RandomVariableHipFactory()
        self._increments = None                 # lazy initialisation (:97)
        self._lock = threading.Lock()

    def getCloneWithModifiedSeed(self, seed):   # :111-113
        return BrownianMotionHip(self.timeDiscretization, self.numberOfFactors, self.numberOfPaths, seed,
                                 self.randomVariableFactory, self.pathOffset)

    def getCloneWithModifiedTimeDiscretization(self, new_time_discretization):   # :116-120
        return BrownianMotionHip(new_time_discretization, self.numberOfFactors, self.numberOfPaths, self.seed,
                                 self.randomVariableFactory, self.pathOffset)

    def getBrownianIncrement(self, time_index, factor):   # :123-139
        with self._lock:
            if self._increments is None:
                self._generate()
        return self._increments[time_index][factor]

    def setGroupSteps(self, steps):
        """Time-step grouping is the engine's business since round 3 (fmhip_set_step_grouping: it watches for the first use of an
        increment with a new time index itself, whatever class hands the increments out); kept for callers of round 2."""
        N.check(N.lib().fmhip_set_step_grouping(int(steps), None))

    def getIncrement(self, time_index, factor):           # :236-238
        return self.getBrownianIncrement(time_index, factor)

    def _generate(self):                                  # doGenerateBrownianMotion, :141-182
        td = self.timeDiscretization
        n_steps = td.getNumberOfTimeSteps()
        dt = np.array([td.getTimeStep(i) for i in range(n_steps)], dtype=np.float64)
        handles = (C.c_int64 * (n_steps * self.numberOfFactors))()
        N.check(N.lib().fmhip_bm_generate(self.seed, n_steps, self.numberOfFactors, self.numberOfPaths,
                                          self.pathOffset, dt.ctypes.data_as(C.POINTER(C.c_double)), handles))
        self._increments = [
            [RandomVariableHip(td.getTime(i + 1), DeviceVector(handles[i * self.numberOfFactors + f], self.numberOfPaths))
             for f in range(self.numberOfFactors)]
            for i in range(n_steps)]

    def getTimeDiscretization(self): return self.timeDiscretization
    def getNumberOfFactors(self): return self.numberOfFactors
    def getNumberOfPaths(self): return self.numberOfPaths
    def getSeed(self): return self.seed
    def getRandomVariableForConstant(self, value): return RandomVariableHip(-float("inf"), value)   # :205-207

    def __eq__(self, o):                                  # :217-233
        return (isinstance(o, BrownianMotionHip) and self.numberOfFactors == o.numberOfFactors
                and self.numberOfPaths == o.numberOfPaths and self.seed == o.seed
                and self.timeDiscretization == o.timeDiscretization)

    def __hash__(self):                                   # :241-247
        r = hash(self.timeDiscretization)
        for v in (self.numberOfFactors, self.numberOfPaths, self.seed):
            r = (31 * r + v) & 0xFFFFFFFF
        return r

    def __repr__(self):
        return (f"BrownianMotionHip(steps={self.timeDiscretization.getNumberOfTimeSteps()}, "
                f"numberOfPaths={self.numberOfPaths}, numberOfFactors={self.numberOfFactors}, seed={self.seed})")


class BrownianMotionFromMersenneRandomNumbers(BrownianMotionHip):
    """finmath-lib's CPU generator (MT19937 + inverse normal CDF), the Brownian motion every reference test uses
    (LIBORMarketModelCalibrationATMTest.java:283, MonteCarloBlackScholesModelTest.java:78-85); increments are generated on
    the host and uploaded through the factory path (double[] → fp32).  Restated from published specifications — the draw
    order is unverified (finmath-lib is not vendored), see csrc/mersenne.cpp."""

    def _generate(self):
        td = self.timeDiscretization
        n_steps = td.getNumberOfTimeSteps()
        dt = np.array([td.getTimeStep(i) for i in range(n_steps)], dtype=np.float64)
        handles = (C.c_int64 * (n_steps * self.numberOfFactors))()
        N.check(N.lib().fmhip_bm_generate_mersenne(self.seed, n_steps, self.numberOfFactors, self.numberOfPaths,
                                                   dt.ctypes.data_as(C.POINTER(C.c_double)), handles))
        self._increments = [
            [RandomVariableHip(td.getTime(i + 1), DeviceVector(handles[i * self.numberOfFactors + f], self.numberOfPaths))
             for f in range(self.numberOfFactors)]
            for i in range(n_steps)]


def mersenne_increments(seed, dt, n_factors, n_paths):
    """Host array [step][factor][path] of the same increments (no device needed)."""
    dt = np.ascontiguousarray(dt, dtype=np.float64)
    out = np.empty((dt.size, n_factors, n_paths), dtype=np.float64)
    N.check(N.lib().fmhip_mersenne_increments(int(seed), dt.size, n_factors, n_paths, dt.ctypes.data_as(C.POINTER(C.c_double)),
                                              out.ctypes.data_as(C.POINTER(C.c_double))))
    return out
