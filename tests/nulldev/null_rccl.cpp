// null_rccl.cpp — TEST-ONLY stand-ins for the RCCL calls of host/lmm_hip_main.cpp (one communicator per process: a world of one) and of
// csrc/sharded.cpp (ONE process, a communicator per device of a device list, the all-gather issued for all of them inside
// ncclGroupStart / ncclGroupEnd), so that the native LMM driver and the device-list front link against the null device
// (tests/nulldev/Makefile) — host-only builds for the sanitizers and for timing the engine's bookkeeping on a machine without a GPU.
#include <rccl/rccl.h>
#include <cstring>
#include <mutex>
#include <vector>
namespace {
struct NullComm { int rank = 0, world = 1; };
struct Pending { const void* send; void* recv; size_t bytes; int rank, world; };
std::mutex g_mu;
int g_group_depth = 0;
std::vector<Pending> g_pending;
void exchange(std::vector<Pending>& calls) {              // every rank's block into every rank's receive buffer, at its rank's offset
    std::vector<std::vector<char>> blocks;
    for (const Pending& p : calls) blocks.emplace_back((const char*)p.send, (const char*)p.send + p.bytes);      // (send may alias recv)
    for (const Pending& dst : calls)
        for (size_t i = 0; i < calls.size(); ++i) std::memcpy((char*)dst.recv + (size_t)calls[i].rank * calls[i].bytes, blocks[i].data(), blocks[i].size());
    calls.clear();
}
}
extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) { std::memset(id, 0, sizeof *id); return ncclSuccess; }
ncclResult_t ncclCommInitRank(ncclComm_t* c, int world, ncclUniqueId, int rank) { *c = reinterpret_cast<ncclComm_t>(new NullComm{ rank, world }); return ncclSuccess; }
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int*) { for (int d = 0; d < ndev; ++d) comms[d] = reinterpret_cast<ncclComm_t>(new NullComm{ d, ndev }); return ncclSuccess; }
ncclResult_t ncclCommDestroy(ncclComm_t c) { delete reinterpret_cast<NullComm*>(c); return ncclSuccess; }
const char* ncclGetErrorString(ncclResult_t) { return "null-rccl error"; }
ncclResult_t ncclGroupStart() { std::lock_guard<std::mutex> lock(g_mu); ++g_group_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd() { std::lock_guard<std::mutex> lock(g_mu); if (--g_group_depth == 0) exchange(g_pending); return ncclSuccess; }
ncclResult_t ncclAllGather(const void* s, void* r, size_t count, ncclDataType_t, ncclComm_t c, hipStream_t) {
    const NullComm* comm = reinterpret_cast<const NullComm*>(c);
    std::lock_guard<std::mutex> lock(g_mu);
    if (comm->world == 1) { if (s != r) std::memmove(r, s, count * 8); return ncclSuccess; }
    g_pending.push_back({ s, r, count * 8, comm->rank, comm->world });
    if (g_group_depth == 0 && (int)g_pending.size() == comm->world) exchange(g_pending);
    return ncclSuccess;
}
}
