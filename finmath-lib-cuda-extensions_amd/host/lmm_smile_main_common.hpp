// lmm_smile_main_common.hpp — command line + JSON report shared by lmm_smile_hip (product) and oracle/host/lmm_smile_cpu (CPU twin).
#pragma once
#include <cstdlib>
#include <string>
#include "lmm_smile.hpp"

namespace fmhost { namespace smile {

struct Options {
    int64_t paths = 8192; int64_t seed = 314151;                       // LIBORMarketModelCalibrationTest.java:72, :267
    int maxIterations = 30; std::string mode = "calibrate"; bool verbose = false, profile = false;
    int evaluations = 1, jacobianBatch = 0;
    bool fullHorizon = true;                                           // all 40 Euler steps, as the reference's Euler scheme simulates them
    std::string brownian = "philox";                                    // philox (on the device) | mersenne (finmath's generator, the one the test injects)
};
inline Options parseOptions(int argc, char** argv) {
    Options o;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(2); } return argv[++i]; };
        if (a == "--paths") o.paths = std::atoll(next());
        else if (a == "--seed") o.seed = std::atoll(next());
        else if (a == "--max-iterations") o.maxIterations = std::atoi(next());
        else if (a == "--mode") o.mode = next();                      // calibrate | evaluate | selftest
        else if (a == "--evaluations") o.evaluations = std::atoi(next());
        else if (a == "--jacobian-batch") o.jacobianBatch = std::atoi(next());
        else if (a == "--brownian") o.brownian = next();
        else if (a == "--lazy-horizon") o.fullHorizon = false;        // stop at the last exercise date (20 steps instead of 40; same results)
        else if (a == "--profile") o.profile = true;
        else if (a == "--verbose") o.verbose = true;
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); std::exit(2); }
    }
    return o;
}

inline void printVolatilities(const std::vector<double>& v) {
    for (size_t k = 0; k < v.size(); ++k) { if (v[k] != v[k]) std::printf("%snull", k ? ", " : ""); else std::printf("%s%.17g", k ? ", " : "", v[k]); }
}

// mode evaluate: `evaluations` objective evaluations at the initial parameters, --jacobian-batch K of them in lock-step
inline void runAndReport(const Options& o, const Backend& be, const char* backendName, const std::function<std::string()>& extraJson) {
    Market m;
    if (o.mode == "selftest") {                                       // host-side pieces, no simulation: numbers for tests/test_lmm_smile_cpu.py
        std::printf("{\"par_swaprate\": [");
        for (size_t k = 0; k < m.products.size(); ++k) std::printf("%s%.17g", k ? ", " : "", m.products[k].parSwaprate);
        std::printf("], \"annuity\": [");
        for (size_t k = 0; k < m.products.size(); ++k) std::printf("%s%.17g", k ? ", " : "", m.products[k].annuity);
        std::printf("], \"valid\": [");
        for (size_t k = 0; k < m.products.size(); ++k) std::printf("%s%d", k ? ", " : "", m.products[k].valid ? 1 : 0);
        // factor reduction at the initial decay and one finite-difference bump further
        const int n = m.numberOfLibors(), F = CovarianceModel::FACTORS;
        for (int bump = 0; bump < 2; ++bump) {
            const std::vector<double> f = reducedCorrelationFactors(n, F, 0.10 + bump * 1e-4, m.periodLength);
            std::printf("], \"%s\": [", bump ? "factors_bumped" : "factors");
            for (size_t k = 0; k < f.size(); ++k) std::printf("%s%.17g", k ? ", " : "", f[k]);
        }
        // implied volatility of Black values (round trip) for the smile strikes
        std::printf("], \"implied_round_trip\": [");
        bool first = true;
        for (const Product& p : m.products)
            for (double vol : { 0.2, 0.3, 0.559, 1.5 }) {
                const double value = blackValue(p.parSwaprate, vol, p.exercise, p.swaprate, p.annuity);
                std::printf("%s[%.17g, %.17g]", first ? "" : ", ", vol, blackImpliedVolatility(p.parSwaprate, p.exercise, p.swaprate, p.annuity, value));
                first = false;
            }
        std::printf("]}\n");
        return;
    }
    if (o.mode == "evaluate") {
        Valuation v;
        double sim = 0, val = 0;
        const int K = std::max(1, o.jacobianBatch);
        for (int k = 0; k < o.evaluations; k += K) {
            const std::vector<Valuation> vs = evaluateMany(m, std::vector<Parameters>((size_t)std::min(K, o.evaluations - k), initialParameters()), be, o.fullHorizon);
            for (const Valuation& x : vs) { sim += x.seconds_simulation; val += x.seconds_valuation; }
            v = vs.back();
        }
        std::printf("{\"backend\": \"%s\", \"mode\": \"evaluate\", \"paths\": %lld, \"products\": %zu, \"evaluations\": %d, "
                    "\"seconds_simulation_per_evaluation\": %.6f, \"seconds_valuation_per_evaluation\": %.6f, \"launches_per_evaluation\": %lld%s, \"model_volatility\": [",
                    backendName, (long long)o.paths, m.products.size(), o.evaluations, sim / o.evaluations, val / o.evaluations, v.launches, extraJson().c_str());
        printVolatilities(v.modelVolatility);
        std::printf("]}\n");
        return;
    }
    const CalibrationResult r = calibrate(m, be, o.maxIterations, 1e-6, 0.1, 1e-4, o.fullHorizon, o.verbose);
    int valid = 0; for (const Product& p : m.products) valid += p.valid;
    std::printf("{\"backend\": \"%s\", \"mode\": \"calibrate\", \"paths\": %lld, \"products\": %zu, \"products_valued\": %d, \"parameters_calibrated\": 8, "
                "\"iterations\": %d, \"accepted_points\": %d, \"evaluations\": %d, \"speculative_evaluations_discarded\": %d, \"seconds\": %.4f, \"seconds_simulation\": %.4f, \"seconds_valuation\": %.4f, "
                "\"initial_rms\": %.6e, \"rms_deviation\": %.6e, \"mean_deviation\": %.6e%s, \"parameters\": {",
                backendName, (long long)o.paths, m.products.size(), valid, r.iterations, r.accepted, r.evaluations, r.speculative_discarded, r.seconds, r.seconds_simulation, r.seconds_valuation,
                r.initialRms, r.rmsDeviation, r.meanDeviation, extraJson().c_str());
    for (int k = 0; k < 8; ++k) std::printf("%s\"%s\": %.10g", k ? ", " : "", parameterName(k), r.parameters[(size_t)k]);
    std::printf("}, \"model_volatility\": [");
    printVolatilities(r.modelVolatility);
    std::printf("]}\n");
}

}} // namespace fmhost::smile
