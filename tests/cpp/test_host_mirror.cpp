// test_host_mirror.cpp — the reference's RandomVariableGPUTest restated for the C++ host mirror (fmhost::RandomVariableHip)
// against the CPU twin (fmhost::RandomVariableFromFloatArray over the C oracle), through the same interface.
// Known answers: RandomVariableGPUTest.java:69-188; differential operator test: :191-360 (enforced bit-exactly here).
// Built and run by tests/test_gpu_cpp_mirror.py.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <random>
#include <string>
#include <vector>
#include "../../finmath-lib-cuda-extensions_amd/host/random_variable.hpp"
#include "../../oracle/host/random_variable_cpu.hpp"

using namespace fmhost;
static int failures = 0;
#define EXPECT(cond, what) do { if (!(cond)) { std::printf("FAIL %s (%s:%d)\n", what, __FILE__, __LINE__); ++failures; } } while (0)

static bool sameBits(const std::vector<double>& a, const std::vector<double>& b, bool libm) {
    if (a.size() != b.size()) return false;
    size_t diff = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        if (a[i] != a[i] && b[i] != b[i]) continue;
        if (a[i] == b[i]) continue;
        if (!libm) return false;
        if (std::fabs(a[i] - b[i]) > 1e-7 * (1 + std::fabs(b[i]))) return false;     // RandomVariableGPUTest.java:217
        ++diff;
    }
    return diff <= a.size() / 10000;
}

static void knownAnswers(const RandomVariableFactory& f, const char* name) {
    RV rv = f.createRandomVariable(2.0);                                             // :69-86
    rv = rv->mult(2.0)->add(1.0)->squared()->sub(4.0)->div(7.0);
    EXPECT(rv->getAverage() == 3.0 && rv->getVariance() == 0.0, name);
    rv = f.createRandomVariable(0.0, std::vector<double>{ -4.0, -2.0, 0.0, 2.0, 4.0 });  // :89-122
    rv = rv->add(4.0)->div(2.0)->mult(2.0)->div(2.0);
    EXPECT(std::fabs(rv->getAverage() - 2.0) <= 1e-7 && rv->getVariance() == 2.0, name);
    RV rv2 = f.createRandomVariable(3.0)->mult(rv);
    EXPECT(rv2->getAverage() == 6.0 && rv2->getVariance() == 18.0, name);
    for (int size : { 2, 3, 4, 5, 7, 10, 13, 99, 100, 1000, 1024, 2047, 2048, 2049, 20000, 200000 }) {   // :125-153
        std::vector<double> v((size_t)size);
        for (int i = 0; i < size; ++i) v[(size_t)i] = i;
        const double want = size * (size - 1.0) / 2.0 / size;
        EXPECT(std::fabs(f.createRandomVariable(0.0, v)->getAverage() - want) <= want * 1e-6, name);
    }
    RV x = f.createRandomVariable(0.0, std::vector<double>{ 3.0, 1.0, 0.0, 2.0, 4.0, 1.0 / 3.0 });      // :156-188
    RV c1 = x->sqrt()->sub(x->pow(0.5)), c2 = x->squared()->sub(x->pow(2.0));
    EXPECT(std::fabs(c1->getAverage()) <= 1e-7 && std::fabs(c1->getVariance()) <= 1e-7, name);
    EXPECT(std::fabs(c2->getAverage()) <= 1e-7 && std::fabs(c2->getVariance()) <= 1e-7, name);
    EXPECT(std::fabs(std::sqrt(x->getVariance()) - x->getStandardDeviation()) <= 1e-7, name);
}

int main() {
    check(fmhip_init(-1));
    RandomVariableHipFactory hip;
    RandomVariableFloatFactory cpu;
    knownAnswers(hip, "known answers (hip)");
    knownAnswers(cpu, "known answers (cpu twin)");

    const int n = 100000;                                                            // :194-201
    std::vector<double> stream((size_t)n);
    orc_java_random_doubles(31415, n, stream.data());
    const double third = 1.0 / 3.0;
    using F = std::function<RV(const RV&, const RV&)>;
    struct Case { const char* name; F f; bool libm; };
    const std::vector<Case> cases = {
        { "squared", [](const RV& x, const RV&) { return x->squared(); }, false },
        { "add_s", [&](const RV& x, const RV&) { return x->add(third); }, false },
        { "add_xy", [](const RV& x, const RV& y) { return x->add(y); }, false }, { "add_yx", [](const RV& x, const RV& y) { return y->add(x); }, false },
        { "sub_xy", [](const RV& x, const RV& y) { return x->sub(y); }, false }, { "sub_yx", [](const RV& x, const RV& y) { return y->sub(x); }, false },
        { "bus_xy", [](const RV& x, const RV& y) { return x->bus(y); }, false }, { "bus_yx", [](const RV& x, const RV& y) { return y->bus(x); }, false },
        { "bus_xx", [](const RV& x, const RV&) { return x->bus(x->mult(2.0)); }, false },
        { "cap_s", [&](const RV& x, const RV&) { return x->cap(third); }, false }, { "cap_yx", [](const RV& x, const RV& y) { return y->cap(x); }, false },
        { "cap_xy", [](const RV& x, const RV& y) { return x->cap(y); }, false },
        { "floor_s", [&](const RV& x, const RV&) { return x->floor(third); }, false }, { "floor_yx", [](const RV& x, const RV& y) { return y->floor(x); }, false },
        { "floor_xx", [](const RV& x, const RV&) { return x->floor(x->squared()); }, false },
        { "mult_xx", [](const RV& x, const RV&) { return x->mult(x); }, false }, { "mult_yx", [](const RV& x, const RV& y) { return y->mult(x); }, false },
        { "mult_s", [](const RV& x, const RV&) { return x->mult(3.1415); }, false },
        { "div_xx", [](const RV& x, const RV&) { return x->div(x->add(1.0)); }, false }, { "div_yx", [](const RV& x, const RV& y) { return y->div(x); }, false },
        { "div_xy", [](const RV& x, const RV& y) { return x->div(y); }, false }, { "div_s", [](const RV& x, const RV&) { return x->div(3.1415); }, false },
        { "vid_xy", [](const RV& x, const RV& y) { return x->vid(y->cap(0.75)->floor(0.75)); }, false }   /* fp32-valued constant: DESIGN.md §2, vid(constant) */, { "vid_yx", [](const RV& x, const RV& y) { return y->vid(x); }, false },
        { "vid_s", [](const RV& x, const RV&) { return x->vid(2.0); }, false }, { "bus_s", [](const RV& x, const RV&) { return x->bus(2.0); }, false },
        { "exp", [](const RV& x, const RV&) { return x->exp(); }, true }, { "log", [](const RV& x, const RV&) { return x->log(); }, true },
        { "pow", [](const RV& x, const RV&) { return x->pow(1.5); }, true },
        { "sqrt", [](const RV& x, const RV&) { return x->sqrt(); }, false }, { "invert", [](const RV& x, const RV&) { return x->invert(); }, false },
        { "abs", [](const RV& x, const RV&) { return x->sub(0.5)->abs(); }, false }, { "isNaN", [](const RV& x, const RV&) { return x->sub(0.5)->sqrt()->isNaN(); }, false },
        { "accrue_xx", [](const RV& x, const RV&) { return x->accrue(x, 2.0); }, false }, { "accrue_xy", [&](const RV& x, const RV& y) { return x->accrue(y, third); }, false },
        { "accrue_yx", [&](const RV& x, const RV& y) { return y->accrue(x, third); }, false },
        { "discount_xx", [](const RV& x, const RV&) { return x->discount(x, 2.0); }, false }, { "discount_xy", [&](const RV& x, const RV& y) { return x->discount(y, third); }, false },
        { "discount_yx", [&](const RV& x, const RV& y) { return y->discount(x, third); }, false },
        { "addProduct_xxx", [](const RV& x, const RV&) { return x->addProduct(x, x); }, false }, { "addProduct_xxy", [](const RV& x, const RV& y) { return x->addProduct(x, y); }, false },
        { "addProduct_xyx", [](const RV& x, const RV& y) { return x->addProduct(y, x); }, false }, { "addProduct_yxx", [](const RV& x, const RV& y) { return y->addProduct(x, x); }, false },
        { "addProduct_yxy", [](const RV& x, const RV& y) { return y->addProduct(x, y); }, false },
        { "addProduct_s_xx", [&](const RV& x, const RV&) { return x->addProduct(x, third); }, false }, { "addProduct_s_yx", [&](const RV& x, const RV& y) { return y->addProduct(x, third); }, false },
        { "addProduct_s_xy", [&](const RV& x, const RV& y) { return x->addProduct(y, third); }, false },
        { "choose", [](const RV& x, const RV& y) { return x->sub(0.5)->choose(x, y); }, false }, { "choose_det", [](const RV& x, const RV& y) { return y->choose(x, x->squared()); }, false },
        { "addRatio", [](const RV& x, const RV&) { return x->addRatio(x->add(1.0), x->add(2.0)); }, false }, { "subRatio", [](const RV& x, const RV& y) { return x->subRatio(y, x->add(2.0)); }, false },
        { "sin", [](const RV& x, const RV&) { return x->sin(); }, true }, { "cos", [](const RV& x, const RV&) { return x->mult(3.0)->cos(); }, true },
        { "apply", [](const RV& x, const RV&) { return x->apply([](double v) { return v * v + 1.0 / 3.0; })->add(1.0); }, false },
        { "chain", [](const RV& x, const RV& y) { return x->add(4.0)->div(2.0)->mult(y)->sub(x)->squared()->cap(9.0)->floor(0.25)->addProduct(x, y)->discount(x, 0.5); }, false },
    };
    for (int fusion = 0; fusion <= 1; ++fusion) {
        check(fmhip_set_fusion(fusion, nullptr));
        for (const Case& c : cases) {
            const RV xh = hip.createRandomVariable(0.0, stream), yh = hip.createRandomVariable(0.0, stream[0]);
            const RV xc = cpu.createRandomVariable(0.0, stream), yc = cpu.createRandomVariable(0.0, stream[0]);
            const RV rh = c.f(xh, yh), rc = c.f(xc, yc);
            const bool ok = sameBits(rh->getRealizations(), rc->getRealizations(), c.libm) && rh->getFiltrationTime() == rc->getFiltrationTime();
            if (!ok) { std::printf("FAIL case %s (fusion %d)\n", c.name, fusion); ++failures; }
        }
        // mixed types: the higher type priority takes over, the foreign operand is uploaded (RandomVariableCuda.java:759-766, :1392)
        const RV xh = hip.createRandomVariable(2.0, stream), xc = cpu.createRandomVariable(1.0, stream);
        const RV r1 = xh->add(xc), r2 = xc->add(xh), want = xc->add(xc);
        EXPECT(dynamic_cast<const RandomVariableHip*>(r1.get()) && dynamic_cast<const RandomVariableHip*>(r2.get()), "type priority");
        EXPECT(sameBits(r1->getRealizations(), want->getRealizations(), false) && sameBits(r2->getRealizations(), want->getRealizations(), false), "foreign upload");
        EXPECT(r1->getFiltrationTime() == 2.0 && r2->getFiltrationTime() == 2.0, "filtration time");
    }
    check(fmhip_set_fusion(0, nullptr));
    // reductions
    {
        const RV xh = hip.createRandomVariable(0.0, stream), xc = cpu.createRandomVariable(0.0, stream);
        EXPECT(std::fabs(xh->getAverage() - xc->getAverage()) <= 1e-13, "average");
        EXPECT(std::fabs(xh->getVariance() - xc->getVariance()) <= 1e-13, "variance");
        EXPECT(xh->getMin() == xc->getMin() && xh->getMax() == xc->getMax(), "min/max");
        EXPECT(std::fabs(xh->getStandardError() - xc->getStandardError()) <= 1e-13, "standard error");
        bool threw = false;
        try { xh->doubleValue(); } catch (const UnsupportedOperation&) { threw = true; }
        EXPECT(threw, "doubleValue on stochastic throws");
        const std::vector<double> av = getAverages({ xh, xh->squared(), hip.createRandomVariable(1.5) });
        EXPECT(av[0] == xh->getAverage() && av[1] == xh->squared()->getAverage() && av[2] == 1.5, "getAverages");
    }
    // differential fuzzing of the dispatch logic: random method chains over constants and stochastic variables with different
    // filtration times, the HIP mirror against the CPU twin class (same idea as tests/test_gpu_fuzz_mirror.py, C++ flavour)
    for (int fusion = 0; fusion <= 1; ++fusion) {
        check(fmhip_set_fusion(fusion, nullptr));
        for (unsigned seed = 0; seed < 60; ++seed) {
            std::mt19937 rng(9000 + seed);
            auto pickInt = [&](int n) { return (int)(rng() % (unsigned)n); };
            const int len = std::vector<int>{ 1, 2, 513, 4099 }[(size_t)pickInt(4)];
            std::vector<double> xa((size_t)len), xb((size_t)len);
            orc_java_random_doubles(700 + seed, len, xa.data()); orc_java_random_doubles(900 + seed, len, xb.data());
            for (int i = 0; i < len; ++i) { xa[(size_t)i] = xa[(size_t)i] * 2.0 - 0.7; xb[(size_t)i] = xb[(size_t)i] * 2.0 - 0.7; }
            std::vector<RV> h = { hip.createRandomVariable(1.0, xa), hip.createRandomVariable(2.5, xb), hip.createRandomVariable(0.5, 0.75), hip.createRandomVariable(3.0, -2.0) };
            std::vector<RV> c = { cpu.createRandomVariable(1.0, xa), cpu.createRandomVariable(2.5, xb), cpu.createRandomVariable(0.5, 0.75), cpu.createRandomVariable(3.0, -2.0) };
            const double scalars[] = { 0.5, 2.0, -1.5, 1.0 / 3.0, 0.0, 3.0 };
            const int steps = 5 + pickInt(30);
            bool ok = true;
            std::string what;
            for (int k = 0; k < steps && ok; ++k) {
                const int kind = pickInt(7), a = pickInt((int)h.size()), b = pickInt((int)h.size()), d = pickInt((int)h.size()), m = pickInt(8);
                const double sc = scalars[pickInt(6)];
                auto apply = [&](const std::vector<RV>& v) -> RV {
                    const RV& x = v[(size_t)a]; const RV& y = v[(size_t)b]; const RV& z = v[(size_t)d];
                    switch (kind) {
                    case 0: switch (m % 5) { case 0: return x->squared(); case 1: return x->sqrt(); case 2: return x->invert(); case 3: return x->abs(); default: return x->isNaN(); }
                    case 1: switch (m) { case 0: return x->cap(sc); case 1: return x->floor(sc); case 2: return x->add(sc); case 3: return x->sub(sc); case 4: return x->bus(sc); case 5: return x->mult(sc); case 6: return x->div(sc); default: return x->vid(sc); }
                    case 2: switch (m) { case 0: return x->cap(y); case 1: return x->floor(y); case 2: return x->add(y); case 3: return x->sub(y); case 4: return x->bus(y); case 5: return x->mult(y); case 6: return x->div(y);
                            default:    // vid with a constant that is not an fp32 value: the one value-level branch where the classes differ
                                if (y->isDeterministic() && !x->isDeterministic() && (double)(float)y->doubleValue() != y->doubleValue()) return x->vid((double)(float)y->doubleValue());
                                return x->vid(y); }
                    case 3: return (m & 1) ? x->accrue(y, sc) : x->discount(y, sc);
                    case 4: return x->addProduct(y, sc);
                    case 5: switch (m % 3) { case 0: return x->addProduct(y, z); case 1: return x->addRatio(y, z); default: return x->subRatio(y, z); }
                    default: return x->choose(y, z);
                    }
                };
                const RV rh = apply(h), rc = apply(c);
                what = "kind " + std::to_string(kind) + " m " + std::to_string(m) + " (" + std::to_string(a) + "," + std::to_string(b) + "," + std::to_string(d) + ")";
                ok = rh->isDeterministic() == rc->isDeterministic() && rh->getFiltrationTime() == rc->getFiltrationTime()
                     && sameBits(rh->getRealizations(), rc->getRealizations(), false);
                h.push_back(rh); c.push_back(rc);
            }
            if (!ok) { std::printf("FAIL fuzz seed %u fusion %d at %s\n", seed, fusion, what.c_str()); ++failures; }
        }
    }
    check(fmhip_set_fusion(0, nullptr));
    // host-side cold paths (sort on the host, both back ends through the same interface code)
    {
        const RV xh = hip.createRandomVariable(0.0, stream), xc = cpu.createRandomVariable(0.0, stream);
        for (double q : { 0.0, 0.01, 0.25, 0.5, 0.99, 1.0 }) EXPECT(xh->getQuantile(q) == xc->getQuantile(q), "getQuantile");
        EXPECT(std::fabs(xh->getQuantile(0.25) - 0.75) < 0.01, "getQuantile uses 1 - quantile (RandomVariableCuda.java:983)");
        EXPECT(xh->getQuantileExpectation(0.1, 0.6) == xc->getQuantileExpectation(0.1, 0.6) && xh->getQuantileExpectation(0.6, 0.1) == xh->getQuantileExpectation(0.1, 0.6), "getQuantileExpectation");
        EXPECT(std::fabs(xh->getQuantileExpectation(0.0, 1.0) - xh->getAverage()) < 1e-9, "full-range quantile expectation = average");
        const std::vector<double> pts = { 0.1, 0.5, 0.9 };
        const std::vector<double> hh = xh->getHistogram(pts), hc = xc->getHistogram(pts);
        EXPECT(hh == hc && hh.size() == 4, "getHistogram");
        EXPECT(std::fabs(hh[0] - 0.1) < 0.01 && std::fabs(hh[1] - 0.4) < 0.01 && std::fabs(hh[3] - 0.1) < 0.01, "histogram of a uniform sample");
        const auto h2 = xh->getHistogram(5, 2.0);
        EXPECT(h2.size() == 2 && h2[0].size() == 6 && h2[1].size() == 6, "getHistogram(points, sd)");
        double total = 0.0; for (double v : h2[1]) total += v;
        EXPECT(std::fabs(total - 1.0) < 1e-12, "histogram sums to one");
        const RV det = hip.createRandomVariable(2.5);
        EXPECT(det->getQuantile(0.3) == 2.5 && det->getQuantileExpectation(0.1, 0.9) == 2.5 && det->sin()->doubleValue() == std::sin(2.5)
               && det->apply([](double v) { return 2 * v; })->doubleValue() == 5.0, "deterministic cold paths");
    }
    // Brownian motion: identical increments on both back ends
    {
        TimeDiscretization td(0.0, 3, 0.25);
        BrownianMotionHip bh(td, 2, 4097, 1234);
        BrownianMotionCpu bc(td, 2, 4097, 1234);
        for (int t = 0; t < 3; ++t) for (int f = 0; f < 2; ++f)
            EXPECT(sameBits(bh.getBrownianIncrement(t, f)->getRealizations(), bc.getBrownianIncrement(t, f)->getRealizations(), false)
                   && bh.getBrownianIncrement(t, f)->getFiltrationTime() == td.getTime(t + 1), "brownian increment");
    }
    {   // clones and equality (BrownianMotionCudaWithRandomVariableCuda.java:131-139, :230-259)
        TimeDiscretization td(0.0, 2, 0.5);
        BrownianMotionHip b(td, 1, 1001, 42);
        auto same = b.getCloneWithModifiedTimeDiscretization(td), other = b.getCloneWithModifiedSeed(43);
        EXPECT(b == *same && !(b == *other) && other->getSeed() == 43, "clone / equals");
        EXPECT(sameBits(b.getIncrement(1, 0)->getRealizations(), same->getBrownianIncrement(1, 0)->getRealizations(), false), "clone reproduces the increments");
        EXPECT(!sameBits(b.getIncrement(1, 0)->getRealizations(), other->getBrownianIncrement(1, 0)->getRealizations(), false), "another seed, other increments");
    }
    check(fmhip_pool_purge());
    std::printf("%s: %zu operator cases x 2 modes, %d failures\n", failures ? "FAILED" : "OK", cases.size(), failures);
    return failures ? 1 : 0;
}
