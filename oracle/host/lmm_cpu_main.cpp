// lmm_cpu — the same LMM driver on the CPU twin (TEST INFRASTRUCTURE: parity checks and the cpu_baseline of the LMM
// benchmark).  Single-threaded, one fresh array per method call (the reference's CPU cost model).
#include "random_variable_cpu.hpp"
#include "../../finmath-lib-cuda-extensions_amd/host/lmm_main_common.hpp"
using namespace fmhost;
int main(int argc, char** argv) {
    const lmm::Options o = lmm::parseOptions(argc, argv);
    try {
        RandomVariableFloatFactory factory;
        lmm::Market m;
        BrownianMotionCpu philox(m.timeDiscretization, 1, o.paths, o.seed, o.pathOffset);
        BrownianMotionFromMersenneRandomNumbers mersenne(m.timeDiscretization, 1, o.paths, (int)o.seed, &factory);
        lmm::Backend be;
        be.factory = &factory;
        be.brownianMotion = (o.brownian == "mersenne") ? static_cast<const BrownianMotion*>(&mersenne) : &philox;
        lmm::runAndReport(o, be, "cpu-twin", [] { return std::string(", \"cores\": 1"); });
    } catch (const std::exception& e) { std::fprintf(stderr, "lmm_cpu: %s\n", e.what()); return 1; }
    return 0;
}
