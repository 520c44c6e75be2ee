// lmm_smile_hip — the reference's swaption smile calibration (LIBORMarketModelCalibrationTest.java) on the MI355X engine:
// 5-factor LMM with blended local volatility and stochastic volatility, 19 swaptions quoted in log-normal volatility,
// Levenberg–Marquardt over 8 parameters (host/lmm_smile.hpp).  One process = one GPU.  The only workload the reference
// publishes wall times for (README.md:232-257).
#include "lmm_smile_main_common.hpp"
#include "hip_backend.hpp"
using namespace fmhost;

int main(int argc, char** argv) {
    const smile::Options o = smile::parseOptions(argc, argv);
    try {
        check(fmhip_init(-1));
        check(fmhip_set_fusion(1, nullptr));
        RandomVariableHipFactory factory;
        smile::Market m;
        BrownianMotionHip philox(m.timeDiscretization, 6, o.paths, o.seed, 0);                                       // :267: 5 factors + 1 for the volatility
        BrownianMotionFromMersenneRandomNumbers mersenne(m.timeDiscretization, 6, o.paths, (int)o.seed, &factory);   // drawn on the host, uploaded through the factory
        lmm::Backend be = makeHipBackend(&factory, (o.brownian == "mersenne") ? static_cast<const BrownianMotion*>(&mersenne) : &philox, 0, 0, o.jacobianBatch);
        fmhip_pool_stats_t s0; check(fmhip_pool_stats(&s0));
        int64_t bytes0 = 0; check(fmhip_traffic_stats(&bytes0, nullptr));
        if (o.profile) check(fmhip_profile_enable(1));
        auto extra = [&] {
            fmhip_pool_stats_t s; check(fmhip_pool_stats(&s));
            char name[128] = { 0 }; int cus = 0; int64_t hbm = 0; fmhip_device_info(name, 128, &cus, &hbm);
            int64_t jc = 0, jf = 0, jp = 0, jd = 0; double js = 0.0; fmhip_jit_stats(&jc, &jf, &jp, &js, &jd);
            int64_t bytes1 = 0, jl = 0; fmhip_traffic_stats(&bytes1, &jl);
            std::string prof;
            if (o.profile) {
                double ms = 0.0; int64_t nl = 0; fmhip_profile_read(&ms, &nl);
                char pb[256];
                std::snprintf(pb, sizeof pb, ", \"profiled_launches\": %lld, \"kernel_ms_total\": %.3f, \"achieved_GBps\": %.1f",
                              (long long)nl, ms, ms > 0 ? (double)(bytes1 - bytes0) / (ms * 1e-3) / 1e9 : 0.0);
                prof = pb;
            }
            char buf[640];
            std::snprintf(buf, sizeof buf, ", \"kernel_launches\": %lld, \"path_ops\": %.6e, \"device_bytes_reserved\": %lld, \"device\": \"%s\", "
                          "\"specialised_kernels\": %lld, \"specialisations_from_disk_cache\": %lld, \"specialisation_seconds\": %.3f, \"algorithmic_bytes\": %lld, \"specialised_launches\": %lld",
                          (long long)(s.n_kernel_launches - s0.n_kernel_launches), (double)(s.n_ops_executed - s0.n_ops_executed) * (double)o.paths,
                          (long long)s.bytes_reserved, name, (long long)jc, (long long)jd, js, (long long)(bytes1 - bytes0), (long long)jl);
            return std::string(buf) + prof;
        };
        smile::runAndReport(o, be, "hip", extra);
        check(fmhip_shutdown());
    } catch (const std::exception& e) { std::fprintf(stderr, "lmm_smile_hip: %s\n", e.what()); return 1; }
    return 0;
}
