// lmm_hip — LMM ATM swaption calibration on the MI355X engine (BASELINE.json configs[3] and [4]).
// One process = one GPU.  With --world W --rank R --nccl-id-file F the Monte-Carlo paths are sharded over W processes
// (rank R simulates global paths [R·paths, (R+1)·paths) — the counter-based generator makes the union identical to a
// single-GPU run); the ONLY communication is one RCCL all-reduce of the 144 expectation sums per objective evaluation
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
        be.jacobianBatch = o.jacobianBatch > 0 ? o.jacobianBatch : 8;   // default: 8 finite-difference bumps in lock-step (≈ 13 GB of state each at 1 M paths)
        be.launches = [] { fmhip_pool_stats_t s; check(fmhip_pool_stats(&s)); return (long long)s.n_kernel_launches; };
        be.averages = [](const std::vector<RV>& v) { return getAverages(v); };

        ncclComm_t comm = nullptr;
        fmhip_vec sums = 0;                              // device buffer of the expectation partials (count x 4 doubles)
        long long collectives = 0;
        if (o.world > 1 || !o.ncclIdFile.empty()) {
            if (o.ncclIdFile.empty()) throw std::runtime_error("--world needs --nccl-id-file");
            ncclUniqueId id;
            if (o.rank == 0) {                           // bootstrap: rank 0 publishes the unique id through a file
                ncclCheck(ncclGetUniqueId(&id), "ncclGetUniqueId");
                { std::ofstream f(o.ncclIdFile + ".tmp", std::ios::binary); f.write((const char*)&id, sizeof id); }
                std::rename((o.ncclIdFile + ".tmp").c_str(), o.ncclIdFile.c_str());
            } else {
                for (int tries = 0; ; ++tries) {
                    std::ifstream f(o.ncclIdFile, std::ios::binary);
                    if (f && f.read((char*)&id, sizeof id) && f.gcount() == (std::streamsize)sizeof id) break;
                    if (tries > 6000) throw std::runtime_error("timed out waiting for " + o.ncclIdFile);
                    std::this_thread::sleep_for(std::chrono::milliseconds(10));
                }
            }
            ncclCheck(ncclCommInitRank(&comm, o.world, id, o.rank), "ncclCommInitRank");
            const int count = (int)m.swaptions.size();
            check(fmhip_vec_create_uninitialized((int64_t)count * 8, &sums));       // 32 bytes per product
            void* stream = nullptr; check(fmhip_get_stream(&stream));
            const int64_t totalPaths = (int64_t)o.world * o.paths;
            be.averages = [=, &collectives](const std::vector<RV>& v) {
                std::vector<fmhip_vec> h;
                for (const RV& x : v) {
                    auto p = dynamic_cast<const RandomVariableHip*>(x.get());
                    if (!p || p->isDeterministic()) throw std::runtime_error("sharded expectation of a non-device value");
                    h.push_back(p->deviceVector().handle());
                }
                void* dev = nullptr; check(fmhip_vec_device_ptr(sums, &dev));
                check(fmhip_reduce_moments_batch_device(h.data(), (int)h.size(), nullptr, dev));
                ncclCheck(ncclAllReduce(dev, dev, h.size() * 4, ncclDouble, ncclSum, comm, (hipStream_t)stream), "ncclAllReduce");
                ++collectives;
                std::vector<float> raw(h.size() * 8);
                check(fmhip_vec_read_float(sums, raw.data(), (int64_t)raw.size()));  // same stream: ordered after the all-reduce
                std::vector<double> out(h.size());
                for (size_t k = 0; k < h.size(); ++k) { double s; std::memcpy(&s, &raw[k * 8], 8); out[k] = s / (double)totalPaths; }
                return out;
            };
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
                          "\"world\": %d, \"rank\": %d, \"total_paths\": %lld, \"rccl_all_reduces\": %lld, "
                          "\"specialised_kernels\": %lld, \"specialisations_from_disk_cache\": %lld, \"specialisations_pending\": %lld, \"specialisation_seconds\": %.3f, "
                          "\"algorithmic_bytes\": %lld, \"specialised_launches\": %lld",
                          (long long)(s.n_kernel_launches - s0.n_kernel_launches), (double)(s.n_ops_executed - s0.n_ops_executed) * (double)o.paths,
                          (long long)s.bytes_reserved, name, o.world, o.rank, (long long)o.world * (long long)o.paths, collectives,
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
