// lmm.hpp — native LIBOR-Market-Model Monte-Carlo calibration driver (BASELINE.json configs[3..4]).
//
// The reference runs this workload through finmath-lib (external jar, NOT vendored): LIBORMarketModelCalibrationATMTest
// builds a 1-factor LMM on a 0…40y / 0.5y grid (80 forward rates), piecewise-constant volatility on an 8×8
// (simulation time × time to maturity) grid, exponential-decay correlation reduced to ONE factor, spot measure, normal
// state space, Euler scheme; calibrates the volatility parameters to ATM normal swaption volatilities with
// Levenberg–Marquardt and finite differences, every objective evaluation re-simulating the model and valuing each
// SwaptionSimple by Monte-Carlo → getAverage() (LIBORMarketModelCalibrationATMTest.java:151-470).
//
// What is restated here and what pins it:
//   - inputs: swaption grid and normal vols (:188-236), swap curve (:527-532), grids (:272-278), vol/corr model
//     (:287-288), measure/state space (:307-311), LM settings (:314-340), acceptance |mean deviation| < 2e-4 (:466);
//   - model algebra: textbook LMM (spot-measure drift via running factor sums, Euler step in the normal state space,
//     rolled-over numeraire, SwaptionSimple backward induction, Bachelier ATM inversion).  finmath-lib's exact op order,
//     day-count/schedule conventions and its optimizer's step rules are [unverified: not under /root/reference];
//     idealised year fractions are used.  The op STREAM (which RandomVariable methods run on which vectors) has the
//     same shape, which is what the engine is measured on.
// The driver only uses the fmhost interfaces, so the same code runs on RandomVariableHip and on the CPU twin.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <atomic>
#include <thread>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "random_variable.hpp"

namespace fmhost { namespace lmm {

// ------------------------------------------------------------------ market data (LIBORMarketModelCalibrationATMTest.java)

inline const std::vector<double>& atmNormalVolatilities() {           // :217-236, 14 expiries × 14 tenors
    static const std::vector<double> v = {
        0.00151, 0.00169, 0.0021, 0.00248, 0.00291, 0.00329, 0.00365, 0.004, 0.00437, 0.00466, 0.00527, 0.00571,
        0.00604, 0.00625, 0.0016, 0.00174, 0.00217, 0.00264, 0.00314, 0.00355, 0.00398, 0.00433, 0.00469,
        0.00493, 0.00569, 0.00607, 0.00627, 0.00645, 0.00182, 0.00204, 0.00238, 0.00286, 0.00339, 0.00384,
        0.00424, 0.00456, 0.00488, 0.0052, 0.0059, 0.00623, 0.0064, 0.00654, 0.00205, 0.00235, 0.00272, 0.0032,
        0.00368, 0.00406, 0.00447, 0.00484, 0.00515, 0.00544, 0.00602, 0.00629, 0.0064, 0.00646, 0.00279,
        0.00319, 0.0036, 0.00396, 0.00436, 0.00469, 0.00503, 0.0053, 0.00557, 0.00582, 0.00616, 0.00628,
        0.00638, 0.00641, 0.00379, 0.00406, 0.00439, 0.00472, 0.00504, 0.00532, 0.0056, 0.00582, 0.00602,
        0.00617, 0.0063, 0.00636, 0.00638, 0.00639, 0.00471, 0.00489, 0.00511, 0.00539, 0.00563, 0.00583, 0.006,
        0.00618, 0.0063, 0.00644, 0.00641, 0.00638, 0.00635, 0.00634, 0.00544, 0.00557, 0.00572, 0.00591,
        0.00604, 0.00617, 0.0063, 0.00641, 0.00651, 0.00661, 0.00645, 0.00634, 0.00627, 0.00624, 0.00625,
        0.00632, 0.00638, 0.00644, 0.0065, 0.00655, 0.00661, 0.00667, 0.00672, 0.00673, 0.00634, 0.00614,
        0.00599, 0.00593, 0.00664, 0.00671, 0.00675, 0.00676, 0.00676, 0.00675, 0.00676, 0.00674, 0.00672,
        0.00669, 0.00616, 0.00586, 0.00569, 0.00558, 0.00647, 0.00651, 0.00651, 0.00651, 0.00652, 0.00649,
        0.00645, 0.0064, 0.00637, 0.00631, 0.00576, 0.00534, 0.00512, 0.00495, 0.00615, 0.0062, 0.00618,
        0.00613, 0.0061, 0.00607, 0.00602, 0.00596, 0.00591, 0.00586, 0.00536, 0.00491, 0.00469, 0.0045,
        0.00578, 0.00583, 0.00579, 0.00574, 0.00567, 0.00562, 0.00556, 0.00549, 0.00545, 0.00538, 0.00493,
        0.00453, 0.00435, 0.0042, 0.00542, 0.00547, 0.00539, 0.00532, 0.00522, 0.00516, 0.0051, 0.00504, 0.005,
        0.00495, 0.00454, 0.00418, 0.00404, 0.00394 };
    return v;
}
inline const std::vector<double>& atmExpiryYears() { static const std::vector<double> v = { 1.0 / 12, 0.25, 0.5, 1, 2, 3, 4, 5, 7, 10, 15, 20, 25, 30 }; return v; }
inline const std::vector<double>& atmTenorYears()  { static const std::vector<double> v = { 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 15, 20, 25, 30 }; return v; }

// Discount curve: log-linear interpolation of discount factors, constant extrapolation (:609-618), bootstrapped from the
// par swap rates of :527-532 (annual fixed leg against the single curve; idealised year fractions).
struct DiscountCurve {
    std::vector<double> t{ 0.0 }, logdf{ 0.0 };
    double df(double time) const {
        if (time <= 0.0) return 1.0;
        if (time >= t.back()) return std::exp(logdf.back());
        const size_t k = (size_t)(std::upper_bound(t.begin(), t.end(), time) - t.begin());
        const double w = (time - t[k - 1]) / (t[k] - t[k - 1]);
        return std::exp(logdf[k - 1] + w * (logdf[k] - logdf[k - 1]));
    }
    double forward(double fixing, double period) const { return (df(fixing) / df(fixing + period) - 1.0) / period; }
    static DiscountCurve bootstrap() {
        const double maturities[] = { 0.5, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 15, 20, 25, 30, 35, 40, 45, 50 };
        const double rates[] = { -0.00216, -0.00208, -0.00222, -0.00216, -0.0019, -0.0014, -0.00072, 0.00011, 0.00103, 0.00196, 0.00285,
                                 0.00367, 0.0044, 0.00604, 0.00733, 0.00767, 0.00773, 0.00765, 0.00752, 0.007138, 0.007 };
        DiscountCurve c;
        for (int i = 0; i < 21; ++i) {
            const double M = maturities[i], r = rates[i];
            c.t.push_back(M); c.logdf.push_back(c.logdf.back());
            auto swapValue = [&](double ld) {                       // receiver-float par swap: 1 - DF(M) - r·annuity
                c.logdf.back() = ld;
                double annuity = 0.0;
                if (M < 1.0) annuity = M * c.df(M);
                else for (double tp = 1.0; tp <= M + 1e-9; tp += 1.0) annuity += c.df(tp);
                return 1.0 - c.df(M) - r * annuity;
            };
            double lo = -3.0, hi = 1.0;                              // value is decreasing in log DF(M)
            for (int it = 0; it < 200; ++it) { const double mid = 0.5 * (lo + hi); if (swapValue(mid) > 0) lo = mid; else hi = mid; }
            c.logdf.back() = 0.5 * (lo + hi);
        }
        return c;
    }
};

struct Swaption {                       // createCalibrationItem (:475-520): ATM, VOLATILITYNORMAL, SwaptionSimple
    double exercise = 0; int numberOfPeriods = 0; double swapPeriodLength = 0.5;
    std::vector<double> swapTenor; double swaprate = 0, targetVolatility = 0, annuity = 0;
};

struct Market {
    DiscountCurve curve = DiscountCurve::bootstrap();
    double lastTime = 40.0, dt = 0.5;                                // :275-277
    TimeDiscretization timeDiscretization{ 0.0, 80, 0.5 };           // simulation grid = LIBOR period grid (:278)
    std::vector<Swaption> swaptions;
    Market() {
        const auto& vols = atmNormalVolatilities();
        for (size_t e = 0; e < atmExpiryYears().size(); ++e)
            for (size_t k = 0; k < atmTenorYears().size(); ++k) {
                const double exercise = std::round(atmExpiryYears()[e] / 0.25) * 0.25, tenor = atmTenorYears()[k];
                if (exercise < 1.0) continue;                        // :252-254
                // a swap reaching beyond the 40y LIBOR grid cannot be valued; finmath's calibration silently drops such
                // products (the exception is swallowed) — they are excluded explicitly here
                if (exercise + tenor > lastTime + 1e-9) continue;
                Swaption s;
                s.exercise = exercise; s.numberOfPeriods = (int)std::lround(tenor / 0.5);
                for (int p = 0; p <= s.numberOfPeriods; ++p) s.swapTenor.push_back(exercise + p * 0.5);
                double floating = 0.0, annuity = 0.0;                // Swap.getForwardSwapRate (:665-667)
                for (int p = 0; p < s.numberOfPeriods; ++p) {
                    const double d = curve.df(s.swapTenor[p + 1]);
                    floating += curve.forward(s.swapTenor[p], 0.5) * 0.5 * d; annuity += 0.5 * d;
                }
                s.swaprate = floating / annuity; s.annuity = annuity;
                s.targetVolatility = vols[e * atmTenorYears().size() + k];
                swaptions.push_back(std::move(s));
            }
    }
    int numberOfLibors() const { return timeDiscretization.getNumberOfTimeSteps(); }
};

// LIBORVolatilityModelPiecewiseConstant on the grid {0,1,2,5,10,20,30,40}² with initial value 0.5 % (:287)
struct VolatilityModel {
    std::vector<double> grid{ 0.0, 1.0, 2.0, 5.0, 10.0, 20.0, 30.0, 40.0 };
    std::vector<double> parameter = std::vector<double>(64, 0.50 / 100);
    int bin(double x) const { int k = 0; while (k < 7 && grid[(size_t)k] < x - 1e-12) ++k; return k; }
    int parameterIndex(double time, double maturity) const { return bin(time) * 8 + bin(maturity - time); }
    double volatility(double time, double maturity) const { return maturity <= time + 1e-12 ? 0.0 : parameter[(size_t)parameterIndex(time, maturity)]; }
    // parameters that some (simulation time, LIBOR) pair actually reads
    std::vector<int> activeParameters(const Market& m) const {
        std::vector<char> used(64, 0);
        const auto& td = m.timeDiscretization;
        for (int i = 0; i < td.getNumberOfTimeSteps(); ++i)
            for (int j = i + 1; j < m.numberOfLibors(); ++j) used[(size_t)parameterIndex(td.getTime(i), td.getTime(j))] = 1;
        std::vector<int> idx;
        for (int k = 0; k < 64; ++k) if (used[(size_t)k]) idx.push_back(k);
        return idx;
    }
};

// ------------------------------------------------------------------ simulation (Euler, spot measure, normal state space)

struct Simulation {
    std::vector<std::vector<RV>> libor;     // [time index][component]
    std::vector<RV> numeraire;              // [time index]
};

struct Backend {
    const RandomVariableFactory* factory = nullptr;
    const BrownianMotion* brownianMotion = nullptr;
    std::function<void()> flush = [] {};    // executes pending (lazily fused) work; no-op on an eager back end
    // hold(true): a lazily fusing back end stops executing pending chains on its own accord until hold(false) + flush(), so
    // that the chains recorded in between — the 144 products x K parameter sets of a valuation — are batched as rows of the
    // same launches instead of running one by one as each passes the engine's size threshold (fmhip_fusion_hold)
    std::function<void(bool)> hold = [](bool) {};
    std::function<long long()> launches = [] { return 0LL; };   // kernel launches so far (statistics only)
    // all Monte-Carlo expectations of one objective evaluation (default: one getAverage() per product)
    std::function<std::vector<double>(const std::vector<RV>&)> averages = [](const std::vector<RV>& v) {
        std::vector<double> a; for (const RV& x : v) a.push_back(x->getAverage()); return a; };
    // Optional: ENQUEUE the expectations of one parameter set and return a function that waits for them.  With it the Jacobian
    // batches of a Levenberg–Marquardt iteration are pipelined — the host records batch b+1 while the device still works on
    // batch b, and reads b's expectations afterwards — instead of idling the device at every batch boundary.
    std::function<std::function<std::vector<double>()>(const std::vector<RV>&)> averagesAsync;
    // true: averagesAsync executes whatever is still pending below the values it is given (and may take their expectations in the
    // launches that compute them) — the driver then does not flush between recording the products and asking for their expectations
    bool expectationsRunPending = false;
    // Optional: tells the back end that only the expectations of these values will be asked for, never the values themselves (a lazily
    // fusing back end then need not write them to memory: fmhip_vec_give_up_values).  Called for ALL parameter sets of a batch before
    // the first expectation is enqueued: the launches carry the rows of every set.
    std::function<void(const std::vector<RV>&)> valuesNotNeeded;
    int chunk = 7;                          // components per multi-output launch (≤ 8 outputs incl. the running sum); used when stepsPerLaunch == 1
    // Euler steps recorded back to back before the engine is asked to execute (hold + one flush per group).  The engine schedules
    // the pending graph of the group component by component (runtime.cpp: build_big, consumers first), finds that the schedule is
    // periodic — the same operations for one component after another, the running factor sums carried along — and runs the stretch
    // as ONE launch of a rolled-loop kernel: the state between the steps of a group never touches HBM, every component is read once
    // and written once per GROUP.  Until that kernel is compiled (and for the ragged first components) the graph runs as launches
    // of a few components x all steps of the group.  A group never spans a time index whose state a product reads (`keep`).
    // 1 = one step at a time, flushed every `chunk` components (the scheme of round 1).  The arithmetic per path is the same
    // either way.  4 measured best (2 / 4 / 6 / 8: 3.25 / 3.17–3.25 / 3.23 / 3.22 s per calibration; segmented launches only: 4.0 s).
    int stepsPerLaunch = 4;
    // Optional (lazily fusing back ends): replicate the PENDING expressions below `roots` once per entry of leafTo — copy c reads
    // leafTo[c][i] wherever the original reads leafFrom[i] and takes its scalar operands, in recording order, from (*scalars)[c]
    // (nullptr: the original's) — and return the copies of the roots (fmhip_graph_clone).  With it the parameter sets of a Jacobian
    // batch are not recorded one by one: set 0 is, the others are copies (≈ 50 ns per operation instead of a method call through
    // the mirror classes, a handle and a release).  recordedScalars: the scalar operands below `roots` as recorded, for the
    // driver's check of its own scalar lists.  Results are bit-identical either way.
    std::function<std::vector<std::vector<RV>>(const std::vector<RV>& roots, const std::vector<RV>& leafFrom, const std::vector<std::vector<RV>>& leafTo,
                                               const std::vector<std::vector<double>>* scalars)> clone;
    std::function<std::vector<double>(const std::vector<RV>& roots)> recordedScalars;
    int jacobianBatch = 1;                  // finite-difference bumps evaluated in lock-step (rows of the same launches)
    // true: every time step's state keeps its handle until the evaluation is over, as finmath-lib's Euler scheme stores the whole
    // discretised process (80 x 80 vectors, SURVEY.md §8d config 4: 25.6 GB) — measurement of a caller that knows nothing about the engine
    bool keepAllStates = false;
    int threads = 1;                                       // > 1 (with jacobianBatch 1): the columns of a Jacobian are evaluated by this many threads side by side, as finmath-lib's
                                                           // optimiser does with its thread pool (…ATMTest.java:319); the back end must allow it (fmhip_set_thread_engines)
};

// Simulates SEVERAL parameter sets in lock-step (same Brownian increments = common random numbers): the operations of
// one chunk are recorded for every parameter set before the flush, so that a lazily fusing back end finds K independent
// DAGs of identical structure and runs them as K rows of ONE launch (K times the bytes per launch: at 1 M paths a single
// set's launches are only ~65 MB and launch-granularity-bound).  Each set's arithmetic is unchanged.
// `keep` (optional): time indices whose state a product will read; the state of every other past time step is released as
// soon as the next step exists (one objective evaluation then holds ≈ 15 instead of 80 time steps of 80 vectors).
inline std::vector<Simulation> simulateMany(const Market& m, const std::vector<const VolatilityModel*>& vols, const Backend& be, int lastTimeIndex,
                                            const std::vector<char>* keep = nullptr) {
    const auto& td = m.timeDiscretization;
    const int n = m.numberOfLibors();
    const double delta = m.dt;
    const size_t K = vols.size();
    std::vector<Simulation> sims(K);
    for (Simulation& sim : sims) {
        sim.libor.resize((size_t)lastTimeIndex + 1);
        sim.numeraire.resize((size_t)lastTimeIndex + 1);
        sim.libor[0].resize((size_t)n);
        for (int j = 0; j < n; ++j) sim.libor[0][(size_t)j] = be.factory->createRandomVariable(0.0, m.curve.forward(td.getTime(j), delta));
        sim.numeraire[0] = be.factory->createRandomVariable(0.0, 1.0);
    }
    std::vector<RV> factorSum(K);                                                        // Σ_k λ_k δ/(1+δ L_k): running over components
    const int S = std::max(1, be.stepsPerLaunch);
    static int cloneChecks = 0;                                                          // the first groups verify the scalar lists against the recording
    for (int i0 = 0; i0 < lastTimeIndex;) {
        int i1 = std::min(i0 + S, lastTimeIndex);
        // A state somebody reads (an exercise date) INSIDE a group does not end it (until round 4 it did: the first ten time steps ran as five
        // groups of two): the group keeps that state's handles, so it is one more value the group's launches store — one read of the state
        // less, and one launch of the four-step kernel instead of two of the two-step one.  FMHIP_LMM_GROUPS_END_AT_KEPT=1: the old grouping.
        static const bool endAtKept = std::getenv("FMHIP_LMM_GROUPS_END_AT_KEPT") && std::getenv("FMHIP_LMM_GROUPS_END_AT_KEPT")[0] == '1';
        if (keep && endAtKept) for (int s2 = i0 + 1; s2 < i1; ++s2) if ((*keep)[(size_t)s2]) { i1 = s2; break; }
        // From time index 2 on every value of the state is a vector (L_0 is fixed at time 0, the bank account is constant until
        // L_1 is): the group's pending graph has the same shape for every parameter set — record set 0, replicate the others.
        const bool cloning = K > 1 && S > 1 && keep && be.clone && i0 >= 2;
        const size_t Krec = cloning ? 1 : K;
        if (S > 1) be.hold(true);
        for (int i = i0; i < i1; ++i) {
            const double t = td.getTime(i), dt = td.getTimeStep(i);
            const RV dW = be.brownianMotion->getBrownianIncrement(i, 0);
            for (size_t k = 0; k < Krec; ++k) {
                auto& cur = sims[k].libor[(size_t)i];
                auto& nxt = sims[k].libor[(size_t)i + 1];
                nxt.resize((size_t)n);
                for (int j = 0; j <= i && j < n; ++j) nxt[(size_t)j] = cur[(size_t)j];       // fixed LIBORs
                factorSum[k] = nullptr;
            }
            for (int j0 = i + 1; j0 < n; j0 += be.chunk) {
                const int j1 = std::min(n, j0 + be.chunk);
                for (size_t k = 0; k < Krec; ++k) {
                    auto& cur = sims[k].libor[(size_t)i];
                    auto& nxt = sims[k].libor[(size_t)i + 1];
                    for (int j = j0; j < j1; ++j) {
                        const double lambda = vols[k]->volatility(t, td.getTime(j));         // one factor: loading = volatility
                        // temporaries die before the flush: a live handle would make them extra outputs of the fused launch
                        const RV& L = cur[(size_t)j];
                        const RV transform = be.factory->createRandomVariable(lambda * delta)->discount(L, delta);   // λδ/(1+δL)
                        factorSum[k] = factorSum[k] ? factorSum[k]->add(transform) : transform;
                        const RV drift = factorSum[k]->mult(lambda);
                        nxt[(size_t)j] = L->addProduct(drift, dt)->addProduct(dW, lambda);   // Euler step, normal state space
                    }
                }
                if (S == 1 && j1 - j0 == be.chunk) be.flush();
            }
            for (size_t k = 0; k < Krec; ++k)
                sims[k].numeraire[(size_t)i + 1] = sims[k].numeraire[(size_t)i]->accrue(sims[k].libor[(size_t)i][(size_t)i], delta);   // rolled-over bank account
            for (size_t k = 0; k < Krec; ++k) factorSum[k] = nullptr;
        }
        if (cloning) {
            // roots: the state at the end of the group (components that moved) and the bank account after every step;
            // substituted operands: the state and the bank account at the start; scalars: what the loop above passed, in its order
            std::vector<RV> roots, leafFrom;
            for (int j = i0 + 1; j < n; ++j) roots.push_back(sims[0].libor[(size_t)i1][(size_t)j]);
            for (int i = i0; i < i1; ++i) roots.push_back(sims[0].numeraire[(size_t)i + 1]);
            for (int s2 = i0 + 1; s2 < i1; ++s2)                                             // states inside the group that a product reads: the components alive at that time
                if (!keep || (*keep)[(size_t)s2]) for (int j = s2; j < n; ++j) roots.push_back(sims[0].libor[(size_t)s2][(size_t)j]);   // (no list: every state is kept)
            for (int j = i0; j < n; ++j) leafFrom.push_back(sims[0].libor[(size_t)i0][(size_t)j]);
            leafFrom.push_back(sims[0].numeraire[(size_t)i0]);
            auto scalarsOf = [&](const VolatilityModel& vol) {
                std::vector<double> sc;
                for (int i = i0; i < i1; ++i) {
                    const double t = td.getTime(i), dt = td.getTimeStep(i);
                    for (int j = i + 1; j < n; ++j) {
                        const double lambda = vol.volatility(t, td.getTime(j));
                        // constant.discount(L, δ) = L·δ → +1 → (λδ)/·;  factor sum · λ;  L + drift·dt;  … + dW·λ
                        sc.push_back(delta); sc.push_back(1.0); sc.push_back(lambda * delta); sc.push_back(lambda); sc.push_back(dt); sc.push_back(lambda);
                    }
                    sc.push_back(delta);                                                   // accrue(L_i, δ)
                }
                return sc;
            };
            if (cloneChecks < 64 && be.recordedScalars) {
                ++cloneChecks;
                if (be.recordedScalars(roots) != scalarsOf(*vols[0])) throw std::runtime_error("lmm: the scalar list of the replicated Euler steps does not match the recording");
            }
            std::vector<std::vector<RV>> leafTo(K - 1);
            std::vector<std::vector<double>> scalars(K - 1);
            for (size_t k = 1; k < K; ++k) {
                for (int j = i0; j < n; ++j) leafTo[k - 1].push_back(sims[k].libor[(size_t)i0][(size_t)j]);
                leafTo[k - 1].push_back(sims[k].numeraire[(size_t)i0]);
                scalars[k - 1] = scalarsOf(*vols[k]);
            }
            const std::vector<std::vector<RV>> copies = be.clone(roots, leafFrom, leafTo, &scalars);
            for (size_t k = 1; k < K; ++k) {
                auto& nxt = sims[k].libor[(size_t)i1];
                nxt.resize((size_t)n);
                for (int j = 0; j <= i0 && j < n; ++j) nxt[(size_t)j] = sims[k].libor[(size_t)i0][(size_t)j];     // fixed before the group
                size_t r = 0;
                for (int j = i0 + 1; j < n; ++j) nxt[(size_t)j] = copies[k - 1][r++];
                for (int i = i0; i < i1; ++i) sims[k].numeraire[(size_t)i + 1] = copies[k - 1][r++];
                for (int s2 = i0 + 1; s2 < i1; ++s2)
                    if (!keep || (*keep)[(size_t)s2]) { auto& mid = sims[k].libor[(size_t)s2]; mid.resize((size_t)n); for (int j = s2; j < n; ++j) mid[(size_t)j] = copies[k - 1][r++]; }
            }
        }
        // states no product reads lose their handles BEFORE the flush: a pending value without a handle is an intermediate of the
        // fused launches, not an output (the state between the steps of a group is never materialised)
        if (keep)
            for (int i = i0; i < i1; ++i)
                if (!(*keep)[(size_t)i])
                    for (size_t k = 0; k < K; ++k) { sims[k].libor[(size_t)i].clear(); sims[k].libor[(size_t)i].shrink_to_fit(); }
        if (S > 1) be.hold(false);
        be.flush();
        i0 = i1;
    }
    return sims;
}

inline Simulation simulate(const Market& m, const VolatilityModel& vol, const Backend& be, int lastTimeIndex) {
    return std::move(simulateMany(m, { &vol }, be, lastTimeIndex)[0]);
}

// SwaptionSimple with ValueUnit VOLATILITYNORMAL: backward induction of the swap value at exercise, payoff floored at 0,
// numeraire-relative, Monte-Carlo average, Bachelier inversion (ATM: closed form).
inline RV swaptionValue(const Market& m, const Simulation& sim, const Swaption& s) {
    const auto& td = m.timeDiscretization;
    const int exerciseIndex = td.getTimeIndex(s.exercise);
    RV value;
    for (int p = s.numberOfPeriods - 1; p >= 0; --p) {
        const int j = td.getTimeIndex(s.swapTenor[(size_t)p]);
        const RV& libor = sim.libor[(size_t)exerciseIndex][(size_t)j];
        const RV payoff = libor->sub(s.swaprate)->mult(s.swapPeriodLength);
        value = (value ? value->add(payoff) : payoff)->discount(libor, s.swapPeriodLength);
    }
    return value->floor(0.0)->div(sim.numeraire[(size_t)exerciseIndex]);
}
inline double bachelierAtmImpliedVolatility(double optionValue, double optionMaturity, double annuity) {
    return optionValue * std::sqrt(2.0 * 3.14159265358979323846) / (annuity * std::sqrt(optionMaturity));
}

struct Valuation { std::vector<double> modelVolatility; double seconds_simulation = 0, seconds_valuation = 0; long long launches_simulation = 0, launches_valuation = 0; };

// An objective evaluation of K parameter sets whose expectations have been enqueued but not read yet (Backend::averagesAsync).
struct PendingValuations {
    std::vector<std::function<std::vector<double>()>> expectations;     // per parameter set: waits for the 144 Monte-Carlo averages
    std::vector<Valuation> out;
};

inline std::vector<Valuation> evaluateManyFinish(const Market& m, PendingValuations& p) {
    for (size_t k = 0; k < p.out.size(); ++k) {
        const std::vector<double> optionValues = p.expectations[k]();
        for (size_t q = 0; q < m.swaptions.size(); ++q)
            p.out[k].modelVolatility.push_back(bachelierAtmImpliedVolatility(optionValues[q], m.swaptions[q].exercise, m.swaptions[q].annuity));
    }
    return std::move(p.out);
}

inline PendingValuations evaluateManyBegin(const Market& m, const std::vector<const VolatilityModel*>& vols, const Backend& be) {
    using clk = std::chrono::steady_clock;
    const size_t K = vols.size();
    PendingValuations pending;
    std::vector<Valuation>& out = pending.out;
    out.resize(K);
    int lastIndex = 0;
    for (const Swaption& s : m.swaptions) lastIndex = std::max(lastIndex, m.timeDiscretization.getTimeIndex(s.exercise));
    const auto t0 = clk::now();
    const long long l0 = be.launches();
    std::vector<char> keep((size_t)lastIndex + 1, 0);                // states read by SwaptionSimple: the exercise dates
    for (const Swaption& s : m.swaptions) keep[(size_t)m.timeDiscretization.getTimeIndex(s.exercise)] = 1;
    const std::vector<Simulation> sims = simulateMany(m, vols, be, lastIndex, be.keepAllStates ? nullptr : &keep);
    const auto t1 = clk::now();
    const long long l1 = be.launches();
    std::vector<std::vector<RV>> values(K);
    if (K > 1 && be.clone) {
        // the 144 payoff chains read the kept states and carry the same scalars for every parameter set: record set 0, replicate
        be.hold(true);
        values[0].reserve(m.swaptions.size());
        for (const Swaption& s : m.swaptions) values[0].push_back(swaptionValue(m, sims[0], s));
        std::vector<RV> leafFrom;
        std::vector<std::vector<RV>> leafTo(K - 1);
        for (int e = 0; e <= lastIndex; ++e) {
            if (!keep[(size_t)e]) continue;
            for (size_t k = 0; k < K; ++k) {
                std::vector<RV>& dst = k == 0 ? leafFrom : leafTo[k - 1];
                for (int j = e; j < m.numberOfLibors(); ++j) dst.push_back(sims[k].libor[(size_t)e][(size_t)j]);
                dst.push_back(sims[k].numeraire[(size_t)e]);
            }
        }
        const std::vector<std::vector<RV>> copies = be.clone(values[0], leafFrom, leafTo, nullptr);
        for (size_t k = 1; k < K; ++k) values[k] = copies[k - 1];
        be.hold(false);
        if (!(be.averagesAsync && be.expectationsRunPending)) be.flush();
    } else
    for (size_t k = 0; k < K; ++k) {            // one parameter set at a time: its 144 products are rows enough per launch, and
        be.hold(true);                          // the device starts on set 0 while the host records set 1 (holding all K sets
        values[k].reserve(m.swaptions.size());  // gave the fastest op stream, 5.7 TB/s, but a 3 % slower calibration)
        for (const Swaption& s : m.swaptions) values[k].push_back(swaptionValue(m, sims[k], s));
        be.hold(false);
        if (!(be.averagesAsync && be.expectationsRunPending)) be.flush();
    }
    if (be.valuesNotNeeded) for (size_t k = 0; k < K; ++k) be.valuesNotNeeded(values[k]);
    for (size_t k = 0; k < K; ++k) {
        if (be.averagesAsync) pending.expectations.push_back(be.averagesAsync(values[k]));
        else { const std::vector<double> now = be.averages(values[k]); pending.expectations.push_back([now] { return now; }); }
    }
    const auto t2 = clk::now();
    for (size_t k = 0; k < K; ++k) {            // statistics: the batch's totals, split evenly
        out[k].launches_simulation = (l1 - l0) / (long long)K; out[k].launches_valuation = (be.launches() - l1) / (long long)K;
        out[k].seconds_simulation = std::chrono::duration<double>(t1 - t0).count() / (double)K;
        out[k].seconds_valuation = std::chrono::duration<double>(t2 - t1).count() / (double)K;
    }
    return pending;
}

inline std::vector<Valuation> evaluateMany(const Market& m, const std::vector<const VolatilityModel*>& vols, const Backend& be) {
    PendingValuations p = evaluateManyBegin(m, vols, be);
    return evaluateManyFinish(m, p);
}

inline Valuation evaluate(const Market& m, const VolatilityModel& vol, const Backend& be) {
    return std::move(evaluateMany(m, { &vol }, be)[0]);
}

// ------------------------------------------------------------------ Levenberg–Marquardt with finite differences (:314-340)

struct CalibrationResult {
    VolatilityModel model; int iterations = 0, evaluations = 0;
    double meanDeviation = 0, rmsDeviation = 0, initialRms = 0, seconds = 0, seconds_simulation = 0, seconds_valuation = 0;
    std::vector<double> modelVolatility;
};

inline bool solveSymmetric(std::vector<double> A, std::vector<double> b, int n, std::vector<double>& x) {   // Gaussian elimination, partial pivoting
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r) if (std::fabs(A[(size_t)r * n + c]) > std::fabs(A[(size_t)piv * n + c])) piv = r;
        if (std::fabs(A[(size_t)piv * n + c]) < 1e-300) return false;
        if (piv != c) { for (int k = 0; k < n; ++k) std::swap(A[(size_t)c * n + k], A[(size_t)piv * n + k]); std::swap(b[(size_t)c], b[(size_t)piv]); }
        for (int r = c + 1; r < n; ++r) {
            const double f = A[(size_t)r * n + c] / A[(size_t)c * n + c];
            for (int k = c; k < n; ++k) A[(size_t)r * n + k] -= f * A[(size_t)c * n + k];
            b[(size_t)r] -= f * b[(size_t)c];
        }
    }
    x.assign((size_t)n, 0.0);
    for (int r = n - 1; r >= 0; --r) { double s = b[(size_t)r]; for (int k = r + 1; k < n; ++k) s -= A[(size_t)r * n + k] * x[(size_t)k]; x[(size_t)r] = s / A[(size_t)r * n + r]; }
    return true;
}

inline CalibrationResult calibrate(const Market& m, const Backend& be, int maxIterations = 200, double accuracy = 1e-7,
                                   double lambda = 0.1, double parameterStep = 1e-4, bool verbose = false) {
    using clk = std::chrono::steady_clock;
    const auto start = clk::now();
    CalibrationResult res;
    VolatilityModel vol;
    const std::vector<int> active = vol.activeParameters(m);
    const int np = (int)active.size(), nr = (int)m.swaptions.size();
    auto residuals = [&](const VolatilityModel& v, std::vector<double>& r, std::vector<double>* modelVols) {
        const Valuation val = evaluate(m, v, be);
        res.evaluations++; res.seconds_simulation += val.seconds_simulation; res.seconds_valuation += val.seconds_valuation;
        r.resize((size_t)nr);
        for (int k = 0; k < nr; ++k) r[(size_t)k] = val.modelVolatility[(size_t)k] - m.swaptions[(size_t)k].targetVolatility;
        if (modelVols) *modelVols = val.modelVolatility;
    };
    auto sumsq = [](const std::vector<double>& r) { double s = 0; for (double x : r) s += x * x; return s; };
    std::vector<double> r, rTrial, modelVols;
    residuals(vol, r, &modelVols);
    double err = sumsq(r);
    res.initialRms = std::sqrt(err / nr);
    for (int it = 0; it < maxIterations; ++it) {
        // Jacobian by forward differences: one re-simulation per active parameter (common random numbers)
        std::vector<double> J((size_t)nr * np);
        // The batches are independent of each other: batch b+1 is recorded and enqueued BEFORE the expectations of batch b are
        // read, so the device never waits for the host at a batch boundary (with a back end that has no averagesAsync the
        // expectations are simply computed inside Begin).
        struct InFlight { int a0 = 0, a1 = 0; std::vector<VolatilityModel> bumped; PendingValuations pending; bool valid = false; };
        auto collect = [&](InFlight& f) {
            if (!f.valid) return;
            const std::vector<Valuation> vals = evaluateManyFinish(m, f.pending);
            for (int a = f.a0; a < f.a1; ++a) {
                const Valuation& val = vals[(size_t)(a - f.a0)];
                res.evaluations++; res.seconds_simulation += val.seconds_simulation; res.seconds_valuation += val.seconds_valuation;
                for (int k = 0; k < nr; ++k) J[(size_t)k * np + a] = (val.modelVolatility[(size_t)k] - m.swaptions[(size_t)k].targetVolatility - r[(size_t)k]) / parameterStep;
            }
            f.valid = false;
        };
        if (be.threads > 1 && be.jacobianBatch <= 1) {
            // one column per evaluation, the evaluations dealt out to the threads: nothing is shared but the model's inputs and the Brownian motion
            const auto tj0 = clk::now();
            std::vector<Valuation> column((size_t)np);
            std::vector<std::string> failures((size_t)be.threads);
            std::atomic<int> next{ 0 };
            std::vector<std::thread> pool;
            for (int t = 0; t < be.threads; ++t)
                pool.emplace_back([&, t] {
                    try {
                        for (int a = next.fetch_add(1); a < np; a = next.fetch_add(1)) {
                            VolatilityModel bumped = vol;
                            bumped.parameter[(size_t)active[(size_t)a]] += parameterStep;
                            column[(size_t)a] = evaluate(m, bumped, be);
                        }
                    } catch (const std::exception& e) { failures[(size_t)t] = e.what(); }
                });
            for (std::thread& th : pool) th.join();
            if (std::getenv("FMHIP_LMM_TIMELINE")) std::fprintf(stderr, "[lmm timeline] iteration %d: %d columns on %d threads, %.3f … %.3f ms\n", it, np, be.threads, std::chrono::duration<double, std::milli>(tj0 - start).count(), std::chrono::duration<double, std::milli>(clk::now() - start).count());
            for (const std::string& f : failures) if (!f.empty()) throw std::runtime_error("a Jacobian column failed: " + f);
            for (int a = 0; a < np; ++a) {
                const Valuation& val = column[(size_t)a];
                res.evaluations++; res.seconds_simulation += val.seconds_simulation; res.seconds_valuation += val.seconds_valuation;
                for (int k = 0; k < nr; ++k) J[(size_t)k * np + a] = (val.modelVolatility[(size_t)k] - m.swaptions[(size_t)k].targetVolatility - r[(size_t)k]) / parameterStep;
            }
        } else {
        InFlight previous;
        for (int a0 = 0; a0 < np; a0 += std::max(1, be.jacobianBatch)) {
            const int a1 = std::min(np, a0 + std::max(1, be.jacobianBatch));
            InFlight current;
            current.a0 = a0; current.a1 = a1;
            current.bumped.assign((size_t)(a1 - a0), vol);
            std::vector<const VolatilityModel*> ptrs;
            for (int a = a0; a < a1; ++a) { current.bumped[(size_t)(a - a0)].parameter[(size_t)active[(size_t)a]] += parameterStep; ptrs.push_back(&current.bumped[(size_t)(a - a0)]); }
            static const bool timeline = std::getenv("FMHIP_LMM_TIMELINE") != nullptr;        // stderr: when a batch was recorded, when the one before it was collected
            const auto tb0 = clk::now();
            try { current.pending = evaluateManyBegin(m, ptrs, be); current.valid = true; }
            catch (const std::exception& e) {           // K simultaneous states did not fit the device: one by one instead
                if (ptrs.size() == 1 || std::string(e.what()).find("allocation") == std::string::npos) throw;
                if (verbose) std::fprintf(stderr, "  batch of %zu bumps does not fit (%s): evaluating them one at a time\n", ptrs.size(), e.what());
                be.flush();
                collect(previous);
                for (int a = a0; a < a1; ++a) {
                    const Valuation val = evaluate(m, current.bumped[(size_t)(a - a0)], be);
                    res.evaluations++; res.seconds_simulation += val.seconds_simulation; res.seconds_valuation += val.seconds_valuation;
                    for (int k = 0; k < nr; ++k) J[(size_t)k * np + a] = (val.modelVolatility[(size_t)k] - m.swaptions[(size_t)k].targetVolatility - r[(size_t)k]) / parameterStep;
                }
                continue;
            }
            const auto tb1 = clk::now();
            collect(previous);
            if (timeline) std::fprintf(stderr, "[lmm timeline] iteration %d batch from %d: begin %.3f … %.3f ms, previous collected at %.3f ms\n", it, a0,
                                       std::chrono::duration<double, std::milli>(tb0 - start).count(), std::chrono::duration<double, std::milli>(tb1 - start).count(),
                                       std::chrono::duration<double, std::milli>(clk::now() - start).count());
            previous = std::move(current);
        }
        collect(previous);
        }
        std::vector<double> JtJ((size_t)np * np, 0.0), Jtr((size_t)np, 0.0);
        for (int a = 0; a < np; ++a) {
            for (int b = 0; b < np; ++b) { double s = 0; for (int k = 0; k < nr; ++k) s += J[(size_t)k * np + a] * J[(size_t)k * np + b]; JtJ[(size_t)a * np + b] = s; }
            double s = 0; for (int k = 0; k < nr; ++k) s += J[(size_t)k * np + a] * r[(size_t)k]; Jtr[(size_t)a] = -s;
        }
        bool improved = false;
        for (int attempt = 0; attempt < 8 && !improved; ++attempt) {
            std::vector<double> A = JtJ, step;
            for (int a = 0; a < np; ++a) A[(size_t)a * np + a] += lambda * std::max(JtJ[(size_t)a * np + a], 1e-12);   // Levenberg regularisation
            if (!solveSymmetric(A, Jtr, np, step)) { lambda *= 4; continue; }
            VolatilityModel v = vol;
            for (int a = 0; a < np; ++a) v.parameter[(size_t)active[(size_t)a]] += step[(size_t)a];
            std::vector<double> mv;
            const auto tt0 = clk::now();
            residuals(v, rTrial, &mv);
            static const bool timeline2 = std::getenv("FMHIP_LMM_TIMELINE") != nullptr;
            if (timeline2) std::fprintf(stderr, "[lmm timeline] iteration %d trial %d: %.3f … %.3f ms\n", it, attempt, std::chrono::duration<double, std::milli>(tt0 - start).count(),
                                        std::chrono::duration<double, std::milli>(clk::now() - start).count());
            const double e = sumsq(rTrial);
            if (e < err) { improved = true; const double rel = (err - e) / err; vol = v; r = rTrial; err = e; modelVols = mv; lambda = std::max(lambda / 3.0, 1e-9); if (rel < accuracy) it = maxIterations; }
            else lambda *= 4.0;
        }
        res.iterations++;
        if (verbose) std::fprintf(stderr, "  LM iteration %d: rms %.6e  lambda %.3g  evaluations %d\n", res.iterations, std::sqrt(err / nr), lambda, res.evaluations);
        if (!improved || std::sqrt(err / nr) < accuracy) break;
    }
    res.model = vol; res.modelVolatility = modelVols;
    double sum = 0; for (double x : r) sum += x;
    res.meanDeviation = sum / nr; res.rmsDeviation = std::sqrt(err / nr);
    res.seconds = std::chrono::duration<double>(clk::now() - start).count();
    return res;
}

}} // namespace fmhost::lmm
