// null_rccl.cpp — TEST-ONLY stand-ins for the five RCCL calls of host/lmm_hip_main.cpp, so that the native LMM driver links against the null
// device (tests/nulldev/Makefile: lmm_null) — a host-only build for timing and profiling the engine's bookkeeping on a machine without a GPU.
#include <rccl/rccl.h>
#include <cstring>
extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) { std::memset(id, 0, sizeof *id); return ncclSuccess; }
ncclResult_t ncclCommInitRank(ncclComm_t* c, int, ncclUniqueId, int) { *c = reinterpret_cast<ncclComm_t>(new int(0)); return ncclSuccess; }
ncclResult_t ncclCommDestroy(ncclComm_t c) { delete reinterpret_cast<int*>(c); return ncclSuccess; }
const char* ncclGetErrorString(ncclResult_t) { return "null-rccl error"; }
ncclResult_t ncclAllGather(const void* s, void* r, size_t count, ncclDataType_t, ncclComm_t, hipStream_t) { if (s != r) std::memmove(r, s, count * 8); return ncclSuccess; }
}
