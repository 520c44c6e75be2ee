// small_n_latency.cpp — latency of the reference README's example chain x.add(4).div(2).exp().getAverage()
// against the path count, through the C++ host mirror (no interpreter in the way): the engine against the CPU twin
// (oracle: one loop + one fresh array per method).  The reference puts its own CPU/GPU break-even at about 5 000 paths (README.md:26).
//   g++ -O2 -std=c++17 -ffp-contract=off -I include -o /tmp/lat benchmarks/small_n_latency.cpp -Lfinmath-lib-cuda-extensions_amd/lib -lfmhip -Loracle -lfm_oracle \
//       -Wl,-rpath,$PWD/finmath-lib-cuda-extensions_amd/lib -Wl,-rpath,$PWD/oracle && /tmp/lat
#include <chrono>
#include <cstdio>
#include <vector>
#include "../finmath-lib-cuda-extensions_amd/host/random_variable.hpp"
#include "../oracle/host/random_variable_cpu.hpp"
using namespace fmhost;

int main() {
    check(fmhip_init(-1));
    check(fmhip_set_fusion(1, nullptr));
    RandomVariableHipFactory hip;
    RandomVariableFloatFactory cpu;
    std::printf("[");
    bool first = true;
    for (int64_t n : { 100, 1000, 5000, 20000, 100000, 1000000 }) {
        std::vector<double> x((size_t)n);
        orc_java_random_doubles(31415, n, x.data());
        const RV g = hip.createRandomVariable(0.0, x), c = cpu.createRandomVariable(0.0, x);
        const int reps = n <= 100000 ? 5000 : 500, creps = n <= 20000 ? 2000 : (n <= 100000 ? 200 : 20);
        double a = 0, b = 0;
        for (int i = 0; i < 100; ++i) a = g->add(4.0)->div(2.0)->exp()->getAverage();
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; ++i) a = g->add(4.0)->div(2.0)->exp()->getAverage();
        const double engine_us = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps * 1e6;
        t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < creps; ++i) b = c->add(4.0)->div(2.0)->exp()->getAverage();
        const double cpu_us = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / creps * 1e6;
        std::printf("%s{\"paths\": %lld, \"engine_us\": %.1f, \"cpu_twin_us\": %.1f, \"same_average\": %s}", first ? "" : ", ", (long long)n, engine_us, cpu_us,
                    std::fabs(a - b) <= 1e-12 * std::fabs(b) ? "true" : "false");
        first = false;
    }
    std::printf("]\n");
    check(fmhip_shutdown());
    return 0;
}
