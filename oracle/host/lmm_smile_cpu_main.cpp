// lmm_smile_cpu — the smile calibration driver on the CPU twin (TEST INFRASTRUCTURE: parity checks and CPU timing).
// Single-threaded, one fresh array per method call (the reference's CPU cost model).
#include "random_variable_cpu.hpp"
#include "../../finmath-lib-cuda-extensions_amd/host/lmm_smile_main_common.hpp"
using namespace fmhost;
int main(int argc, char** argv) {
    const smile::Options o = smile::parseOptions(argc, argv);
    try {
        RandomVariableFloatFactory factory;
        smile::Market m;
        BrownianMotionCpu philox(m.timeDiscretization, 6, o.paths, o.seed, 0);
        BrownianMotionFromMersenneRandomNumbers mersenne(m.timeDiscretization, 6, o.paths, (int)o.seed, &factory);
        lmm::Backend be;
        be.factory = &factory;
        be.brownianMotion = (o.brownian == "mersenne") ? static_cast<const BrownianMotion*>(&mersenne) : &philox;
        if (o.jacobianBatch > 0) be.jacobianBatch = o.jacobianBatch;
        smile::runAndReport(o, be, "cpu-twin", [] { return std::string(", \"cores\": 1"); });
    } catch (const std::exception& e) { std::fprintf(stderr, "lmm_smile_cpu: %s\n", e.what()); return 1; }
    return 0;
}
