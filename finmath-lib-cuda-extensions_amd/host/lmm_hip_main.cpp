// lmm_hip — LMM ATM swaption calibration on the MI355X engine (BASELINE.json configs[3]).
// Links libfmhip.so only.  One process = one GPU; `--path-offset` selects this process's block of the global path set.
#include "lmm_main_common.hpp"
using namespace fmhost;
int main(int argc, char** argv) {
    const lmm::Options o = lmm::parseOptions(argc, argv);
    try {
        check(fmhip_init(-1));
        check(fmhip_set_fusion(1, nullptr));            // chains of RandomVariable calls run as fused launches
        RandomVariableHipFactory factory;
        lmm::Market m;
        BrownianMotionHip bm(m.timeDiscretization, 1, o.paths, o.seed, o.pathOffset);
        lmm::Backend be;
        be.factory = &factory; be.brownianMotion = &bm;
        be.flush = [] { check(fmhip_flush()); };
        be.averages = [](const std::vector<RV>& v) { return getAverages(v); };
        be.launches = [] { fmhip_pool_stats_t s; check(fmhip_pool_stats(&s)); return (long long)s.n_kernel_launches; };
        fmhip_pool_stats_t s0; check(fmhip_pool_stats(&s0));
        lmm::runAndReport(o, be, "hip", [&] {
            fmhip_pool_stats_t s; check(fmhip_pool_stats(&s));
            char name[128] = { 0 }; int cus = 0; int64_t hbm = 0; fmhip_device_info(name, 128, &cus, &hbm);
            char buf[512];
            std::snprintf(buf, sizeof buf, ", \"kernel_launches\": %lld, \"path_ops\": %.6e, \"device_bytes_reserved\": %lld, \"device\": \"%s\"",
                          (long long)(s.n_kernel_launches - s0.n_kernel_launches), (double)(s.n_ops_executed - s0.n_ops_executed) * (double)o.paths,
                          (long long)s.bytes_reserved, name);
            return std::string(buf);
        });
        check(fmhip_shutdown());
    } catch (const std::exception& e) { std::fprintf(stderr, "lmm_hip: %s\n", e.what()); return 1; }
    return 0;
}
