// lmm_main_common.hpp — command line + JSON report shared by the two LMM driver executables
// (lmm_hip: product, links libfmhip.so only;  oracle/host/lmm_cpu: CPU twin, test infrastructure / cpu_baseline).
#pragma once
#include <vector>
#include <cstdlib>
#include <cstring>
#include <string>
#include "lmm.hpp"

namespace fmhost { namespace lmm {

struct Options {
    int64_t paths = 10000; int64_t seed = 31415; int maxIterations = 200; std::string mode = "calibrate"; bool verbose = false;
    int64_t pathOffset = 0; int evaluations = 1;
    int warmupEvaluations = 0;                             // mode evaluate: parameter sets evaluated once (in one lock-step batch) BEFORE statistics and profiling start —
                                                           // the first batch meets every graph shape for the first time (plans are written down, kernels compiled)
    int world = 1, rank = 0; std::string ncclIdFile; long long ncclNonce = 0;
    std::vector<int> devices;                              // --devices 0,1,2,…: ONE process drives these devices (fmhip_init_devices), `paths` in all, sharded by path blocks
    int chunk = 0;                                         // LIBOR components per fused launch; 0 = back end default
    int stepsPerLaunch = 0;                                // Euler steps recorded per engine flush; 0 = back end default (4)
    int jacobianBatch = 0;                                 // finite-difference bumps simulated in lock-step (rows of one launch); 0 = back end default
    std::string brownian = "philox";                        // philox (counter-based, on the device) | mersenne (finmath's CPU generator through the factory)
    int threads = 1;                                       // --threads T (with --finmath-like): Jacobian columns on T threads, an engine each
    double releaseLagMs = 0.0;                             // --release-lag MS: handles are released as a JVM would release them — by a collector thread, every MS milliseconds,
    long long releaseLagBytes = 0;                         //   everything dead at that moment (ReleaseLag, random_variable.hpp); --release-lag-bytes B: … or once B bytes of dead wrappers have piled up
    bool finmathLike = false;                              // no hints to the engine: no hold / flush / replication / lock-step batches, all states kept, one getAverage per product
    bool profile = false;                                  // bracket every program launch with HIP events (device time of the op stream)      // path sharding over GPUs: one process per GPU
};
inline Options parseOptions(int argc, char** argv) {
    Options o;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(2); } return argv[++i]; };
        if (a == "--paths") o.paths = std::atoll(next());
        else if (a == "--seed") o.seed = std::atoll(next());
        else if (a == "--max-iterations") o.maxIterations = std::atoi(next());
        else if (a == "--mode") o.mode = next();                  // calibrate | evaluate
        else if (a == "--evaluations") o.evaluations = std::atoi(next());
        else if (a == "--warmup-evaluations") o.warmupEvaluations = std::atoi(next());
        else if (a == "--path-offset") o.pathOffset = std::atoll(next());
        else if (a == "--world") o.world = std::atoi(next());
        else if (a == "--devices") { std::string list = next(); size_t p0 = 0; while (p0 <= list.size()) { const size_t q = list.find(',', p0); o.devices.push_back(std::atoi(list.substr(p0, q == std::string::npos ? std::string::npos : q - p0).c_str())); if (q == std::string::npos) break; p0 = q + 1; } }
        else if (a == "--rank") o.rank = std::atoi(next());
        else if (a == "--nccl-id-file") o.ncclIdFile = next();
        else if (a == "--nccl-nonce") o.ncclNonce = std::atoll(next());
        else if (a == "--profile") o.profile = true;
        else if (a == "--finmath-like") o.finmathLike = true;
        else if (a == "--threads") o.threads = std::atoi(next());
        else if (a == "--release-lag") o.releaseLagMs = std::atof(next());
        else if (a == "--release-lag-bytes") o.releaseLagBytes = std::atoll(next());
        else if (a == "--brownian") o.brownian = next();
        else if (a == "--jacobian-batch") o.jacobianBatch = std::atoi(next());
        else if (a == "--chunk") o.chunk = std::atoi(next());
        else if (a == "--steps-per-launch") o.stepsPerLaunch = std::atoi(next());
        else if (a == "--verbose") o.verbose = true;
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); std::exit(2); }
    }
    return o;
}

// mode evaluate: one objective evaluation (simulate + value all swaptions) with the initial parameters, repeated
// `evaluations` times; prints every model volatility with 17 digits so that two back ends can be compared exactly.
inline void runAndReport(const Options& o, const Backend& be, const char* backendName, const std::function<std::string()>& extraJson) {
    Market m;
    if (o.mode == "evaluate") {
        VolatilityModel vol;
        Valuation v;
        double sim = 0, val = 0;
        // --jacobian-batch K (explicit): `evaluations` parameter sets are evaluated K at a time in lock-step, as the calibration does
        const int K = std::max(1, o.jacobianBatch);
        for (int k = 0; k < o.evaluations; k += K) {
            const int kk = std::min(K, o.evaluations - k);
            const std::vector<Valuation> vs = evaluateMany(m, std::vector<const VolatilityModel*>((size_t)kk, &vol), be);
            for (const Valuation& x : vs) { sim += x.seconds_simulation; val += x.seconds_valuation; }
            v = vs.back();
        }
        std::printf("{\"backend\": \"%s\", \"mode\": \"evaluate\", \"paths\": %lld, \"swaptions\": %zu, \"evaluations\": %d, "
                    "\"seconds_simulation_per_evaluation\": %.6f, \"seconds_valuation_per_evaluation\": %.6f, \"launches_simulation\": %lld, \"launches_valuation\": %lld%s, \"model_volatility\": [",
                    backendName, (long long)o.paths, m.swaptions.size(), o.evaluations, sim / o.evaluations, val / o.evaluations, v.launches_simulation, v.launches_valuation, extraJson().c_str());
        for (size_t k = 0; k < v.modelVolatility.size(); ++k) std::printf("%s%.17g", k ? ", " : "", v.modelVolatility[k]);
        std::printf("]}\n");
        return;
    }
    const CalibrationResult r = calibrate(m, be, o.maxIterations, 1e-7, 0.1, 1e-4, o.verbose);
    std::printf("{\"backend\": \"%s\", \"mode\": \"calibrate\", \"paths\": %lld, \"swaptions\": %zu, \"active_parameters\": %zu, "
                "\"iterations\": %d, \"evaluations\": %d, \"seconds\": %.4f, \"seconds_simulation\": %.4f, \"seconds_valuation\": %.4f, "
                "\"initial_rms\": %.6e, \"rms_deviation\": %.6e, \"mean_deviation\": %.6e%s, \"parameters\": [",
                backendName, (long long)o.paths, m.swaptions.size(), VolatilityModel().activeParameters(m).size(), r.iterations, r.evaluations,
                r.seconds, r.seconds_simulation, r.seconds_valuation, r.initialRms, r.rmsDeviation, r.meanDeviation, extraJson().c_str());
    for (size_t k = 0; k < r.model.parameter.size(); ++k) std::printf("%s%.17g", k ? ", " : "", r.model.parameter[k]);      // (17 digits: ranks compare them bit for bit)
    std::printf("]}\n");
}

}} // namespace fmhost::lmm
