// lmm_hip — LMM ATM swaption calibration on the MI355X engine (BASELINE.json configs[3] and [4]).
// One process = one GPU.  With --world W --rank R --nccl-id-file F the Monte-Carlo paths are sharded over W processes
// (rank R simulates global paths [R·paths, (R+1)·paths) — the counter-based generator makes the union identical to a
// single-GPU run); the ONLY communication is one RCCL all-gather of the 144 x {Σ, Σ², min, max} expectation partials per objective evaluation
// (SURVEY.md §8e).  Every rank then takes the same Levenberg–Marquardt step.
#include <fstream>
#include <thread>
#include <csignal>
#include <execinfo.h>
#include <unistd.h>
#include <rccl/rccl.h>
#include "lmm_main_common.hpp"
#include "hip_backend.hpp"
using namespace fmhost;

static void ncclCheck(ncclResult_t r, const char* what) {
    if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r));
}

// FMHIP_BACKTRACE=1: the call stack of a fatal signal on stderr (abort() from the allocator's consistency checks, a segmentation fault)
static void fatalSignal(int sig) {
    void* frames[64];
    const int n = backtrace(frames, 64);
    const char msg[] = "lmm_hip: fatal signal, call stack:\n";
    (void)!write(2, msg, sizeof msg - 1);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

int main(int argc, char** argv) {
    if (std::getenv("FMHIP_BACKTRACE")) { signal(SIGABRT, fatalSignal); signal(SIGSEGV, fatalSignal); signal(SIGBUS, fatalSignal); }
    const lmm::Options o = lmm::parseOptions(argc, argv);
    try {
        if (!o.devices.empty()) check(fmhip_init_devices(o.devices.data(), (int)o.devices.size()));      // one process, several devices: nothing else in this driver changes
        else check(fmhip_init(-1));                      // device = LOCAL_RANK (torchrun) / FMHIP_DEVICE_INDEX / 0
        check(fmhip_set_fusion(1, nullptr));            // chains of RandomVariable calls run as fused launches
        RandomVariableHipFactory factory;
        lmm::Market m;
        const int64_t pathOffset = o.pathOffset + (int64_t)o.rank * o.paths;
        BrownianMotionHip philox(m.timeDiscretization, 1, o.paths, o.seed, pathOffset);
        // --brownian mersenne: the generator the reference's test injects (…ATMTest.java:283), drawn on the host and uploaded
        // through the factory; a sequential stream, so it cannot be sharded by path offset
        BrownianMotionFromMersenneRandomNumbers mersenne(m.timeDiscretization, 1, o.paths, (int)o.seed, &factory);
        if (o.brownian == "mersenne" && o.world > 1) throw std::runtime_error("--brownian mersenne is a sequential stream: not available with --world > 1");
        lmm::Backend be = makeHipBackend(&factory, (o.brownian == "mersenne") ? static_cast<const BrownianMotion*>(&mersenne) : &philox,
                                         o.chunk, o.stepsPerLaunch, o.jacobianBatch);

        if (o.finmathLike) {
            // What finmath-lib's own classes would do through the Java interface: the Euler scheme and the optimizer call RandomVariable
            // methods and getAverage(), nothing else; every time step of the process stays referenced.
            be.flush = [] {}; be.hold = [](bool) {}; be.clone = nullptr; be.recordedScalars = nullptr; be.averagesAsync = nullptr; be.expectationsRunPending = false;
            be.averages = [](const std::vector<RV>& v) { std::vector<double> a; for (const RV& x : v) a.push_back(x->getAverage()); return a; };
            be.jacobianBatch = 1; be.stepsPerLaunch = 1; be.chunk = 1 << 30; be.keepAllStates = true;
            if (o.threads > 1) {                            // the optimiser's thread pool: an engine per thread (fmhip_set_thread_engines)
                if (!o.devices.empty() || o.world > 1) throw std::runtime_error("--threads: one device, one process");
                check(fmhip_set_thread_engines(1, nullptr));
                be.threads = o.threads;
            }
        } else if (o.threads > 1) throw std::runtime_error("--threads belongs to --finmath-like (the native driver batches its Jacobian columns instead)");
        // --release-lag / --release-lag-bytes: the mirror's handles die when a collector thread says so (a JVM caller's lifetime contract)
        if (o.releaseLagMs > 0.0 || o.releaseLagBytes > 0) ReleaseLag::instance().start(o.releaseLagMs, (size_t)o.releaseLagBytes);
        ncclComm_t comm = nullptr;
        std::vector<std::pair<double*, hipEvent_t>> sparePinned;      // pinned blocks and events of finished sharded expectations, used again
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
            // Sharded and pipelined: reduce → all-gather → copy to pinned memory → event, all enqueued; the expectations are read when the
            // NEXT parameter sets have been enqueued, by waiting for that event only (a blocking read of the gather buffer would wait for
            // everything enqueued since: one in-order stream).  Every flight has a gather buffer, a pinned block and an event of its own.
            struct Flight {
                fmhip_vec dev = 0; double* host = nullptr; hipEvent_t ev = nullptr;
                std::vector<std::pair<double*, hipEvent_t>>* spare = nullptr;          // pinned blocks and events are used again (allocating them costs ≈ 0.2 ms)
                ~Flight() { if (dev) fmhip_vec_release(dev); if (host && ev && spare) spare->push_back({ host, ev }); }
            };
            std::vector<std::pair<double*, hipEvent_t>>* const spare = &sparePinned;     // (all of one size: count x world; freed at the end of main)
            // The sharded reduction wants its partials on the device (the send buffer of the all-gather).  Since round 4 the launches that
            // compute the payoffs take their moments along here too and a one-wave kernel collects them into that buffer
            // (fmhip_reduce_moments_batch_device on pending vectors); FMHIP_LMM_SHARDED_FROM_LAUNCHES=0: flush first, then a reduction launch.
            { const char* e = std::getenv("FMHIP_LMM_SHARDED_FROM_LAUNCHES"); be.expectationsRunPending = !(e && e[0] == '0'); }
            be.averagesAsync = [=, &collectives, &collective_seconds](const std::vector<RV>& v) -> std::function<std::vector<double>()> {
                std::vector<fmhip_vec> h;
                for (const RV& x : v) {
                    auto p = dynamic_cast<const RandomVariableHip*>(x.get());
                    if (!p || p->isDeterministic()) throw std::runtime_error("sharded expectation of a non-device value");
                    h.push_back(p->deviceVector().handle());
                }
                if ((int)h.size() != count) throw std::runtime_error("sharded expectation: unexpected product count");
                auto f = std::make_shared<Flight>();
                const size_t doubles = (size_t)count * 4 * (size_t)world;
                check(fmhip_vec_create_uninitialized((int64_t)doubles * 2, &f->dev));
                f->spare = spare;
                if (!spare->empty()) { f->host = spare->back().first; f->ev = spare->back().second; spare->pop_back(); }
                else if (hipHostMalloc((void**)&f->host, doubles * 8, hipHostMallocDefault) != hipSuccess || hipEventCreateWithFlags(&f->ev, hipEventDisableTiming) != hipSuccess)
                    throw std::runtime_error("pinned memory / event for a sharded expectation");
                void* dev = nullptr; check(fmhip_vec_device_ptr(f->dev, &dev));
                double* all = (double*)dev;
                check(fmhip_reduce_moments_batch_device(h.data(), count, nullptr, all + (size_t)rank * count * 4));
                ncclCheck(ncclAllGather(all + (size_t)rank * count * 4, all, (size_t)count * 4, ncclDouble, comm, (hipStream_t)stream), "ncclAllGather");
                if (hipMemcpyAsync(f->host, all, doubles * 8, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess || hipEventRecord(f->ev, (hipStream_t)stream) != hipSuccess)
                    throw std::runtime_error("enqueueing the read-back of a sharded expectation");
                ++collectives;
                return [f, count, world, totalPaths, &collective_seconds] {
                    const auto t0 = std::chrono::steady_clock::now();
                    if (hipEventSynchronize(f->ev) != hipSuccess) throw std::runtime_error("waiting for a sharded expectation");
                    collective_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();      // (what is left to wait for when the result is needed)
                    std::vector<double> out((size_t)count);
                    for (int k = 0; k < count; ++k) {
                        double total = 0.0;
                        for (int r = 0; r < world; ++r) total += f->host[((size_t)r * count + k) * 4];      // sums added in rank order: the same bits on every rank
                        out[(size_t)k] = total / (double)totalPaths;
                    }
                    return out;
                };
            };
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
            std::string lag;
            if (ReleaseLag::instance().on()) {
                const ReleaseLag::Stats ls = ReleaseLag::instance().stats();
                char lb[384];
                std::snprintf(lb, sizeof lb, ", \"release_lag\": {\"collect_every_ms\": %.3f, \"collect_at_bytes\": %lld, \"handles_queued\": %lld, \"handles_released\": %lld, "
                              "\"collections\": %lld, \"collections_forced_by_out_of_memory\": %lld, \"peak_dead_handles\": %lld}",
                              o.releaseLagMs, o.releaseLagBytes, ls.queued, ls.released, ls.collections, ls.forcedCollections, ls.peakQueue);
                lag = lb;
            }
            fmhip_engine_stats_t es; check(fmhip_engine_stats(&es));
            char eb[768];
            std::snprintf(eb, sizeof eb, ", \"engine\": {\"interpreter_launches\": %lld, \"algorithmic_bytes_written\": %lld, \"values_deferred\": %lld, \"values_deferred_now\": %lld, "
                          "\"values_demanded\": %lld, \"peak_bytes_reserved\": %lld, \"late_releases_while_waiting\": %lld, \"late_releases_at_once\": %lld, \"late_release_seconds\": %.3f, "
                          "\"merged_launches\": %lld, \"merged_chains\": %lld, \"common_rows\": %lld}",
                          (long long)es.interpreter_launches, (long long)es.algorithmic_bytes_written, (long long)es.values_deferred, (long long)es.values_deferred_now,
                          (long long)es.values_demanded, (long long)es.peak_bytes_reserved, (long long)es.late_releases_while_waiting, (long long)es.late_releases_at_once, (double)es.late_release_nanoseconds * 1e-9,
                          (long long)es.merged_launches, (long long)es.merged_chains, (long long)es.common_rows);
            return std::string(buf) + prof + lag + eb;
        };
        if (o.rank == 0) lmm::runAndReport(o, be, "hip", extra);
        else { lmm::Options quiet = o; quiet.verbose = false; lmm::runAndReport(quiet, be, "hip", extra); }        // every rank reports: the launcher compares the parameter vectors
        if (sums) fmhip_vec_release(sums);
        ReleaseLag::instance().stop();                      // (the JVM exits: what is still queued is released)
        be = lmm::Backend();                                // (closures that may still hold a flight)
        for (auto& pe : sparePinned) { (void)hipHostFree(pe.first); (void)hipEventDestroy(pe.second); }
        if (comm) ncclCommDestroy(comm);
        check(fmhip_shutdown());
    } catch (const std::exception& e) { std::fprintf(stderr, "lmm_hip[rank %d]: %s\n", o.rank, e.what()); return 1; }
    return 0;
}
