// hip_backend.hpp — how the native Monte-Carlo drivers (lmm_hip, lmm_smile_hip) talk to the engine beyond the RandomVariable
// interface: flush / hold of the lazy front-end, graph replication, batched and enqueued expectations.  C-ABI calls only.
#pragma once
#include <cstdlib>
#include <cstring>
#include <memory>
#include "lmm.hpp"

namespace fmhost {

// chunk / stepsPerLaunch / jacobianBatch: 0 = the back end's default
inline lmm::Backend makeHipBackend(const RandomVariableFactory* factory, const BrownianMotion* brownianMotion, int chunk = 0, int stepsPerLaunch = 0, int jacobianBatch = 0) {
    lmm::Backend be;
    be.factory = factory;
    be.brownianMotion = brownianMotion;
    // FMHIP_LMM_FLUSH=0 (with FMHIP_LMM_HOLD=0, FMHIP_LMM_CLONE=0, FMHIP_LMM_ASYNC=0, --jacobian-batch 1): a caller that gives the
    // engine no hints at all — what finmath-lib's own Euler scheme and optimizer would look like through the Java interface
    if (!(std::getenv("FMHIP_LMM_FLUSH") && std::getenv("FMHIP_LMM_FLUSH")[0] == '0'))
        be.flush = [] { check(fmhip_flush()); };
    if (!(std::getenv("FMHIP_LMM_HOLD") && std::getenv("FMHIP_LMM_HOLD")[0] == '0'))       // =0: measurement of the effect only
        be.hold = [](bool h) { check(fmhip_fusion_hold(h ? 1 : 0, nullptr)); };
    if (chunk > 0) be.chunk = chunk;
    if (stepsPerLaunch > 0) be.stepsPerLaunch = stepsPerLaunch;
    be.jacobianBatch = jacobianBatch > 0 ? jacobianBatch : 8;   // default: 8 finite-difference bumps in lock-step (≈ 13 GB of state each at 1 M paths)
    if (!(std::getenv("FMHIP_LMM_CLONE") && std::getenv("FMHIP_LMM_CLONE")[0] == '0')) {  // =0: every parameter set recorded by hand (A/B measurement)
        auto handleOf = [](const RV& x) {
            auto p = dynamic_cast<const RandomVariableHip*>(x.get());
            if (!p || p->isDeterministic()) throw std::runtime_error("graph replication over a value that is not a device vector");
            return p->deviceVector().handle();
        };
        be.clone = [handleOf](const std::vector<RV>& roots, const std::vector<RV>& leafFrom, const std::vector<std::vector<RV>>& leafTo,
                              const std::vector<std::vector<double>>* scalars) {
            const int nRoots = (int)roots.size(), nMap = (int)leafFrom.size(), nCopies = (int)leafTo.size();
            std::vector<fmhip_vec> r, lf, lt, out((size_t)nRoots * nCopies);
            for (const RV& x : roots) r.push_back(handleOf(x));
            for (const RV& x : leafFrom) lf.push_back(handleOf(x));
            for (const auto& row : leafTo) { if ((int)row.size() != nMap) throw std::runtime_error("graph replication: ragged operand map"); for (const RV& x : row) lt.push_back(handleOf(x)); }
            std::vector<double> sc;
            int nScalars = 0;
            if (scalars) { nScalars = (int)(*scalars)[0].size(); for (const auto& row : *scalars) { if ((int)row.size() != nScalars) throw std::runtime_error("graph replication: ragged scalar lists"); sc.insert(sc.end(), row.begin(), row.end()); } }
            check(fmhip_graph_clone(r.data(), nRoots, nCopies, lf.data(), lt.data(), nMap, scalars ? sc.data() : nullptr, nScalars, out.data()));
            std::vector<std::vector<RV>> copies((size_t)nCopies);
            for (int c = 0; c < nCopies; ++c)
                for (int k = 0; k < nRoots; ++k)
                    copies[(size_t)c].push_back(RandomVariableHip::of(roots[(size_t)k]->getFiltrationTime(), DeviceVector(out[(size_t)c * nRoots + k]), roots[(size_t)k]->size()));
            return copies;
        };
        be.recordedScalars = [handleOf](const std::vector<RV>& roots) {
            std::vector<fmhip_vec> r;
            for (const RV& x : roots) r.push_back(handleOf(x));
            int n = 0;
            check(fmhip_graph_scalars(r.data(), (int)r.size(), nullptr, 0, &n));
            std::vector<double> sc((size_t)n);
            check(fmhip_graph_scalars(r.data(), (int)r.size(), sc.data(), n, &n));
            return sc;
        };
    }
    // the products' payoffs are wanted for their Monte-Carlo averages only: their values are given up (fmhip.h: fmhip_vec_give_up_values),
    // the launches that compute them take the moments and store nothing (FMHIP_LMM_DISCARD=0: not told, A/B measurement)
    if (!(std::getenv("FMHIP_LMM_DISCARD") && std::getenv("FMHIP_LMM_DISCARD")[0] == '0'))
        be.valuesNotNeeded = [](const std::vector<RV>& v) {
            std::vector<fmhip_vec> h;
            for (const RV& x : v) { auto p = dynamic_cast<const RandomVariableHip*>(x.get()); if (p && !p->isDeterministic()) h.push_back(p->deviceVector().handle()); }
            if (!h.empty()) check(fmhip_vec_give_up_values(h.data(), (int)h.size()));
        };
    be.launches = [] { fmhip_engine_stats_t s; check(fmhip_engine_stats(&s)); return (long long)s.kernel_launches; };      // (counters only: fmhip_pool_stats counts live vectors, i.e. performs queued releases first)
    be.averages = [](const std::vector<RV>& v) { return getAverages(v); };
    if (!(std::getenv("FMHIP_LMM_ASYNC") && std::getenv("FMHIP_LMM_ASYNC")[0] == '0'))    // =0: every batch's expectations read before the next batch is recorded (A/B)
        be.expectationsRunPending = true;
    if (!(std::getenv("FMHIP_LMM_ASYNC") && std::getenv("FMHIP_LMM_ASYNC")[0] == '0'))
        be.averagesAsync = [](const std::vector<RV>& v) -> std::function<std::vector<double>()> {
            std::vector<fmhip_vec> h;
            int64_t n = 0;
            for (const RV& x : v) {
                auto p = dynamic_cast<const RandomVariableHip*>(x.get());
                if (!p || p->isDeterministic()) { const std::vector<double> now = getAverages(v); return [now] { return now; }; }      // not all device vectors: no pipelining
                h.push_back(p->deviceVector().handle()); n = p->size();
            }
            const int count = (int)h.size();
            // enqueued; nobody waits.  The ticket is ended when the expectations are read: that waits for THIS reduction only — the next
            // parameter sets, enqueued in between, keep the device busy meanwhile (a blocking read of a result buffer would wait for
            // them too: one in-order stream).
            fmhip_ticket ticket = 0;
            check(fmhip_reduce_moments_batch_begin(h.data(), count, nullptr, &ticket));
            struct Owner { fmhip_ticket t = 0; int count = 0; bool ended = false;       // never read (an exception in between): the ticket is ended all the same
                           Owner() = default; Owner(const Owner&) = delete; Owner& operator=(const Owner&) = delete;
                           ~Owner() { if (!ended && t) { std::vector<fmhip_moments> drop((size_t)count); (void)fmhip_reduce_moments_batch_end(t, drop.data(), count); } } };
            auto owner = std::make_shared<Owner>();
            owner->t = ticket; owner->count = count;
            return [owner, count, n] {
                std::vector<fmhip_moments> m((size_t)count);
                owner->ended = true;
                check(fmhip_reduce_moments_batch_end(owner->t, m.data(), count));        // the only wait of the evaluation
                std::vector<double> out((size_t)count);
                for (int k = 0; k < count; ++k) out[(size_t)k] = m[(size_t)k].sum / (double)n;
                return out;
            };
        };
    return be;
}

} // namespace fmhost
