// mersenne.hpp — header-only restatement of finmath-lib's Mersenne-Twister Brownian motion, shared by
// libfmhip (csrc/mersenne.cpp → fmhip_mersenne_increments / fmhip_bm_generate_mersenne) and by the C++ host mirror
// (BrownianMotionFromMersenneRandomNumbers below, which — like finmath's class — creates its increments through ANY
// RandomVariableFactory, so the HIP engine and the CPU twin can be fed the very same numbers, as the reference's tests do:
// LIBORMarketModelCalibrationATMTest.java:283, MonteCarloBlackScholesModelTest.java:78-85).
//
// That class lives in finmath-lib 5.1.3 (NOT vendored); restated here from published specifications:
//   - MT19937 (Matsumoto & Nishimura) seeded the way org.apache.commons.math3.random.MersenneTwister(long) does it:
//     finmath's wrapper net.finmath.randomnumbers.MersenneTwister takes a `long seed`, so the test's int seed is widened and
//     commons-math3 runs init_by_array({(int)(seed >>> 32), (int)seed}) after init_genrand(19650218) [unverified against
//     the jar: an `int` overload would use plain init_genrand(seed)], and commons-math3's
//     nextDouble() = ((next(26) << 26) | next(26)) · 2^-52;
//   - the inverse normal CDF by Wichura's algorithm AS 241 (PPND16), which finmath's NormalDistribution uses;
//   - increment = inverseCDF(uniform) · sqrt(dt).
// The draw order (path-major: for path, for time step, for factor) is [unverified: finmath-lib source not available];
// MT19937 and AS 241 are pinned by published known answers (tests/test_mersenne_cpu.py).
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace fmhost {

struct MT19937 {
    uint32_t mt[624]; int mti;
    void init_genrand(uint32_t seed) {
        mt[0] = seed;
        for (mti = 1; mti < 624; ++mti) mt[mti] = 1812433253u * (mt[mti - 1] ^ (mt[mti - 1] >> 30)) + (uint32_t)mti;
    }
    // commons-math3 MersenneTwister.setSeed(int[]) = the reference init_by_array of mt19937ar.c
    void init_by_array(const uint32_t* key, int len) {
        init_genrand(19650218u);
        int i = 1, j = 0;
        for (int k = (624 > len ? 624 : len); k != 0; --k) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
            ++i; ++j;
            if (i >= 624) { mt[0] = mt[623]; i = 1; }
            if (j >= len) j = 0;
        }
        for (int k = 623; k != 0; --k) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
            ++i;
            if (i >= 624) { mt[0] = mt[623]; i = 1; }
        }
        mt[0] = 0x80000000u;
        mti = 624;
    }
    explicit MT19937(int64_t seed) {                            // MersenneTwister(long): setSeed(new int[]{ hi, lo })
        const uint32_t key[2] = { (uint32_t)((uint64_t)seed >> 32), (uint32_t)((uint64_t)seed & 0xffffffffu) };
        init_by_array(key, 2);
    }
    uint32_t next32() {
        if (mti >= 624) {
            static const uint32_t mag01[2] = { 0u, 0x9908b0dfu };
            int kk = 0;
            for (; kk < 624 - 397; ++kk) { const uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu); mt[kk] = mt[kk + 397] ^ (y >> 1) ^ mag01[y & 1u]; }
            for (; kk < 623; ++kk) { const uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu); mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ mag01[y & 1u]; }
            const uint32_t y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
            mt[623] = mt[396] ^ (y >> 1) ^ mag01[y & 1u];
            mti = 0;
        }
        uint32_t y = mt[mti++];
        y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18);
        return y;
    }
    double nextDouble() {                                       // commons-math3 BitsStreamGenerator.nextDouble
        const uint64_t high = (uint64_t)(next32() >> 6) << 26;
        const uint64_t low = (uint64_t)(next32() >> 6);
        return (double)(high | low) * 0x1.0p-52;
    }
};

// Wichura (1988), Algorithm AS 241, PPND16: relative accuracy about 1e-16.
inline double inverseNormalCdf(double p) {
    if (!(p > 0.0 && p < 1.0)) return (p == 0.0) ? -HUGE_VAL : (p == 1.0 ? HUGE_VAL : NAN);
    const double q = p - 0.5;
    if (std::fabs(q) <= 0.425) {
        const double r = 0.180625 - q * q;
        return q * (((((((2.5090809287301226727e+3 * r + 3.3430575583588128105e+4) * r + 6.7265770927008700853e+4) * r + 4.5921953931549871457e+4) * r
                        + 1.3731693765509461125e+4) * r + 1.9715909503065514427e+3) * r + 1.3314166789178437745e+2) * r + 3.3871328727963666080e0)
                 / (((((((5.2264952788528545610e+3 * r + 2.8729085735721942674e+4) * r + 3.9307895800092710610e+4) * r + 2.1213794301586595867e+4) * r
                        + 5.3941960214247511077e+3) * r + 6.8718700749205790830e+2) * r + 4.2313330701600911252e+1) * r + 1.0);
    }
    double r = std::sqrt(-std::log(q < 0 ? p : 1.0 - p));
    double val;
    if (r <= 5.0) {
        r -= 1.6;
        val = (((((((7.74545014278341407640e-4 * r + 2.27238449892691845833e-2) * r + 2.41780725177450611770e-1) * r + 1.27045825245236838258e0) * r
                   + 3.64784832476320460504e0) * r + 5.76949722146069140550e0) * r + 4.63033784615654529590e0) * r + 1.42343711074968357734e0)
            / (((((((1.05075007164441684324e-9 * r + 5.47593808499534494600e-4) * r + 1.51986665636164571966e-2) * r + 1.48103976427480074590e-1) * r
                   + 6.89767334985100004550e-1) * r + 1.67638483018380384940e0) * r + 2.05319162663775882187e0) * r + 1.0);
    } else {
        r -= 5.0;
        val = (((((((2.01033439929228813265e-7 * r + 2.71155556874348757815e-5) * r + 1.24266094738807843860e-3) * r + 2.65321895265761230930e-2) * r
                   + 2.96560571828504891230e-1) * r + 1.78482653991729133580e0) * r + 5.46378491116411436990e0) * r + 6.65790464350110377720e0)
            / (((((((2.04426310338993978564e-15 * r + 1.42151175831644588870e-7) * r + 1.84631831751005468180e-5) * r + 7.86869131145613259100e-4) * r
                   + 1.48753612908506148525e-2) * r + 1.36929880922735805310e-1) * r + 5.99832206555887937690e-1) * r + 1.0);
    }
    return q < 0.0 ? -val : val;
}

// out[(step*n_factors + factor)*n_paths + path], doubles (host).  No device involved.
inline void mersenneIncrements(int32_t seed, int n_steps, int n_factors, int64_t n_paths, const double* dt, double* out) {
    MT19937 mt((int64_t)seed);                                  // the int seed of the finmath constructor, widened
    std::vector<double> sq((size_t)n_steps);
    for (int i = 0; i < n_steps; ++i) sq[(size_t)i] = std::sqrt(dt[i]);
    for (int64_t path = 0; path < n_paths; ++path)
        for (int step = 0; step < n_steps; ++step)
            for (int f = 0; f < n_factors; ++f)
                out[((size_t)step * n_factors + f) * (size_t)n_paths + (size_t)path] = inverseNormalCdf(mt.nextDouble()) * sq[(size_t)step];
}

} // namespace fmhost
