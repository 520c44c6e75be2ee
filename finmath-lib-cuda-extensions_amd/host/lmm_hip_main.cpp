// lmm_hip — LMM ATM swaption calibration on the MI355X engine (BASELINE.json configs[3] and [4]).
// One process = one GPU.  With --world W --rank R --nccl-id-file F the Monte-Carlo paths are sharded over W processes
// (rank R simulates global paths [R·paths, (R+1)·paths) — the counter-based generator makes the union identical to a
// single-GPU run); the ONLY communication is one RCCL all-gather of the 144 x {Σ, Σ², min, max} expectation partials per objective evaluation
// (SURVEY.md §8e).  Every rank then takes the same Levenberg–Marquardt step.
#include <fstream>
#include <thread>
#include <rccl/rccl.h>
#include "lmm_main_common.hpp"
using namespace fmhost;

static void ncclCheck(ncclResult_t r, const char* what) {
    if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r));
}

int main(int argc, char** argv) {
    const lmm::Options o = lmm::parseOptions(argc, argv);
    try {
        check(fmhip_init(-1));                           // device = LOCAL_RANK (torchrun) / FMHIP_DEVICE_INDEX / 0
        check(fmhip_set_fusion(1, nullptr));            // chains of RandomVariable calls run as fused launches
        RandomVariableHipFactory factory;
        lmm::Market m;
        const int64_t pathOffset = o.pathOffset + (int64_t)o.rank * o.paths;
        BrownianMotionHip philox(m.timeDiscretization, 1, o.paths, o.seed, pathOffset);
        // --brownian mersenne: the generator the reference's test injects (…ATMTest.java:283), drawn on the host and uploaded
        // through the factory; a sequential stream, so it cannot be sharded by path offset
        BrownianMotionFromMersenneRandomNumbers mersenne(m.timeDiscretization, 1, o.paths, (int)o.seed, &factory);
        if (o.brownian == "mersenne" && o.world > 1) throw std::runtime_error("--brownian mersenne is a sequential stream: not available with --world > 1");
        lmm::Backend be;
        be.factory = &factory;
        be.brownianMotion = (o.brownian == "mersenne") ? static_cast<const BrownianMotion*>(&mersenne) : &philox;
        be.flush = [] { check(fmhip_flush()); };
        if (!(std::getenv("FMHIP_LMM_HOLD") && std::getenv("FMHIP_LMM_HOLD")[0] == '0'))       // =0: measurement of the effect only
            be.hold = [](bool h) { check(fmhip_fusion_hold(h ? 1 : 0, nullptr)); };
        if (o.chunk > 0) be.chunk = o.chunk;
        if (o.stepsPerLaunch > 0) be.stepsPerLaunch = o.stepsPerLaunch;
        be.jacobianBatch = o.jacobianBatch > 0 ? o.jacobianBatch : 8;   // default: 8 finite-difference bumps in lock-step (≈ 13 GB of state each at 1 M paths)
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
        be.launches = [] { fmhip_pool_stats_t s; check(fmhip_pool_stats(&s)); return (long long)s.n_kernel_launches; };
        be.averages = [](const std::vector<RV>& v) { return getAverages(v); };
        if (!(std::getenv("FMHIP_LMM_ASYNC") && std::getenv("FMHIP_LMM_ASYNC")[0] == '0'))    // =0: every batch's expectations read before the next batch is recorded (A/B)
            be.averagesAsync = [](const std::vector<RV>& v) -> std::function<std::vector<double>()> {
                std::vector<fmhip_vec> h;
                int64_t n = 0;
                for (const RV& x : v) {
                    auto p = dynamic_cast<const RandomVariableHip*>(x.get());
                    if (!p || p->isDeterministic()) { const std::vector<double> now = getAverages(v); return [now] { return now; }; }      // not all device vectors: no pipelining
                    h.push_back(p->deviceVector().handle()); n = p->size();
                }
                const int count = (int)h.size();
                fmhip_vec buf = 0;
                check(fmhip_vec_create_uninitialized((int64_t)count * 8, &buf));     // count x {Σ, Σ², min, max} doubles, written by the reduction launch
                void* dev = nullptr; check(fmhip_vec_device_ptr(buf, &dev));
                check(fmhip_reduce_moments_batch_device(h.data(), count, nullptr, dev));       // enqueued; nobody waits
                auto owner = std::make_shared<DeviceVector>(buf);
                return [owner, count, n] {
                    std::vector<float> raw((size_t)count * 8);
                    check(fmhip_vec_read_float(owner->handle(), raw.data(), (int64_t)raw.size()));     // the only synchronisation of the evaluation
                    std::vector<double> out((size_t)count);
                    for (int k = 0; k < count; ++k) { double s; std::memcpy(&s, &raw[(size_t)k * 8], 8); out[(size_t)k] = s / (double)n; }
                    return out;
                };
            };

        ncclComm_t comm = nullptr;
        fmhip_vec sums = 0;                              // device buffer of the expectation partials (count x 4 doubles)
        long long collectives = 0;
        double collective_seconds = 0.0;                 // host wall time from enqueueing the collective to having its result (latency of the exchange incl. the read-back)
        if (o.world > 1 || !o.ncclIdFile.empty()) {
            if (o.ncclIdFile.empty()) throw std::runtime_error("--world needs --nccl-id-file");
            // Bootstrap: rank 0 publishes {nonce, unique id} through a file.  The nonce (--nccl-nonce, the launcher hands every rank the
            // same fresh value) tells the id of THIS launch from one a crashed or earlier run left at the same path: the other ranks
            // ignore a file that carries another nonce.  Rank 0 removes any stale file first and the new one once the communicator exists.
            struct IdFile { uint64_t magic, nonce; ncclUniqueId id; } rec{};
            const uint64_t MAGIC = 0x464d4e43434c4944ull;       // "FMNCCLID"
            if (o.rank == 0) {
                std::remove(o.ncclIdFile.c_str());
                rec.magic = MAGIC; rec.nonce = (uint64_t)o.ncclNonce;
                ncclCheck(ncclGetUniqueId(&rec.id), "ncclGetUniqueId");
                { std::ofstream f(o.ncclIdFile + ".tmp", std::ios::binary); f.write((const char*)&rec, sizeof rec); }
                std::rename((o.ncclIdFile + ".tmp").c_str(), o.ncclIdFile.c_str());
            } else {
                for (int tries = 0; ; ++tries) {
                    std::ifstream f(o.ncclIdFile, std::ios::binary);
                    if (f && f.read((char*)&rec, sizeof rec) && f.gcount() == (std::streamsize)sizeof rec && rec.magic == MAGIC && rec.nonce == (uint64_t)o.ncclNonce) break;
                    if (tries > 6000) throw std::runtime_error("timed out waiting for " + o.ncclIdFile);
                    std::this_thread::sleep_for(std::chrono::milliseconds(10));
                }
            }
            const ncclUniqueId id = rec.id;
            ncclCheck(ncclCommInitRank(&comm, o.world, id, o.rank), "ncclCommInitRank");
            if (o.rank == 0) std::remove(o.ncclIdFile.c_str());                    // every rank has joined, i.e. has read it
            const int count = (int)m.swaptions.size();
            // [rank][product][Σ, Σ², min, max] doubles: every rank's partial moments, gathered (8 floats = 32 bytes per product and rank)
            check(fmhip_vec_create_uninitialized((int64_t)count * 8 * o.world, &sums));
            void* stream = nullptr; check(fmhip_get_stream(&stream));
            const int64_t totalPaths = (int64_t)o.world * o.paths;
            const int world = o.world, rank = o.rank;
            be.averagesAsync = nullptr;                  // sharded: the gather buffer is shared between evaluations, expectations are read at once
            be.averages = [=, &collectives, &collective_seconds](const std::vector<RV>& v) {
                std::vector<fmhip_vec> h;
                for (const RV& x : v) {
                    auto p = dynamic_cast<const RandomVariableHip*>(x.get());
                    if (!p || p->isDeterministic()) throw std::runtime_error("sharded expectation of a non-device value");
                    h.push_back(p->deviceVector().handle());
                }
                if ((int)h.size() != count) throw std::runtime_error("sharded expectation: unexpected product count");
                void* dev = nullptr; check(fmhip_vec_device_ptr(sums, &dev));
                double* all = (double*)dev;
                check(fmhip_reduce_moments_batch_device(h.data(), count, nullptr, all + (size_t)rank * count * 4));
                // The single exchange of an objective evaluation: every rank's {Σ, Σ², min, max} partials to every rank —
                // an all-gather, so that min / max stay meaningful and the sums are added in RANK ORDER on every rank (bitwise equal
                // results everywhere, whatever algorithm the library picks); 4.6 KB per rank, latency-bound.
                const auto t0 = std::chrono::steady_clock::now();
                ncclCheck(ncclAllGather(all + (size_t)rank * count * 4, all, (size_t)count * 4, ncclDouble, comm, (hipStream_t)stream), "ncclAllGather");
                ++collectives;
                std::vector<float> raw((size_t)count * 8 * world);
                check(fmhip_vec_read_float(sums, raw.data(), (int64_t)raw.size()));  // same stream: ordered after the collective
                collective_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                std::vector<double> out((size_t)count);
                for (int k = 0; k < count; ++k) {
                    double total = 0.0;
                    for (int r = 0; r < world; ++r) { double s; std::memcpy(&s, &raw[((size_t)r * count + k) * 8], 8); total += s; }
                    out[(size_t)k] = total / (double)totalPaths;
                }
                return out;
            };
        }

        if (o.mode == "evaluate" && o.warmupEvaluations > 0) {       // untimed, unprofiled: what a calibration does in its first iteration
            lmm::VolatilityModel vol;
            (void)lmm::evaluateMany(m, std::vector<const lmm::VolatilityModel*>((size_t)o.warmupEvaluations, &vol), be);
            check(fmhip_jit_wait());
        }
        fmhip_pool_stats_t s0; check(fmhip_pool_stats(&s0));
        int64_t bytes0 = 0; check(fmhip_traffic_stats(&bytes0, nullptr));
        if (o.profile) check(fmhip_profile_enable(1));
        auto extra = [&] {
            fmhip_pool_stats_t s; check(fmhip_pool_stats(&s));
            char name[128] = { 0 }; int cus = 0; int64_t hbm = 0; fmhip_device_info(name, 128, &cus, &hbm);
            int64_t jc = 0, jf = 0, jp = 0, jd = 0; double js = 0.0; fmhip_jit_stats(&jc, &jf, &jp, &js, &jd);
            int64_t bytes1 = 0, jl = 0; fmhip_traffic_stats(&bytes1, &jl);
            std::string prof;
            if (o.profile) {                             // device time of every program launch of the run
                double ms = 0.0; int64_t nl = 0; fmhip_profile_read(&ms, &nl);
                char pb[256];
                std::snprintf(pb, sizeof pb, ", \"profiled_launches\": %lld, \"kernel_ms_total\": %.3f, \"achieved_GBps\": %.1f",
                              (long long)nl, ms, ms > 0 ? (double)(bytes1 - bytes0) / (ms * 1e-3) / 1e9 : 0.0);
                prof = pb;
            }
            char buf[768];
            std::snprintf(buf, sizeof buf, ", \"kernel_launches\": %lld, \"path_ops\": %.6e, \"device_bytes_reserved\": %lld, \"device\": \"%s\", "
                          "\"world\": %d, \"rank\": %d, \"total_paths\": %lld, \"rccl_collectives\": %lld, \"rccl_collective_seconds\": %.6f, "
                          "\"specialised_kernels\": %lld, \"specialisations_from_disk_cache\": %lld, \"specialisations_pending\": %lld, \"specialisation_seconds\": %.3f, "
                          "\"algorithmic_bytes\": %lld, \"specialised_launches\": %lld",
                          (long long)(s.n_kernel_launches - s0.n_kernel_launches), (double)(s.n_ops_executed - s0.n_ops_executed) * (double)o.paths,
                          (long long)s.bytes_reserved, name, o.world, o.rank, (long long)o.world * (long long)o.paths, collectives, collective_seconds,
                          (long long)jc, (long long)jd, (long long)jp, js, (long long)(bytes1 - bytes0), (long long)jl);
            return std::string(buf) + prof;
        };
        if (o.rank == 0) lmm::runAndReport(o, be, "hip", extra);
        else { lmm::Options quiet = o; quiet.verbose = false; std::FILE* devnull = std::freopen("/dev/null", "w", stdout); (void)devnull; lmm::runAndReport(quiet, be, "hip", extra); }
        if (sums) fmhip_vec_release(sums);
        if (comm) ncclCommDestroy(comm);
        check(fmhip_shutdown());
    } catch (const std::exception& e) { std::fprintf(stderr, "lmm_hip[rank %d]: %s\n", o.rank, e.what()); return 1; }
    return 0;
}
