// null_hip.cpp — TEST-ONLY stand-in for the HIP runtime, hiprtc and the kernel launchers of kernels.hip, so that the host runtime of the
// engine (csrc/runtime.cpp, abi.cpp, jit.cpp: 4 000 lines of hand-managed node pools, intrusive reference counts, replica
// descriptions, tickets, a pinned arena and plan caches) can run under AddressSanitizer, UBSan and ThreadSanitizer on a machine
// without a GPU (GPU sanitizers do not exist on this pool).  It is NOT a back end and never ships: "device" memory is host memory,
// copies are memcpy, launches compute NOTHING — they touch the first and last element of every vector a program launch is handed
// (so a wild or undersized pointer is an ASan report), write placeholder moments where a kernel would have written results, and raise
// the completion flag; events complete at once; hiprtc "compiles" a source into a 16-byte blob that remembers which argument block
// its kernels take.  Built by tests/nulldev/Makefile, driven by tests/nulldev/drive.cpp, run by tests/test_sanitizers_cpu.py.
#include <hip/hip_runtime_api.h>
#include <hip/hiprtc.h>
#include <ctime>
#include <cstdlib>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <algorithm>
#include <unordered_map>

#include "../../finmath-lib-cuda-extensions_amd/csrc/kernels.h"

namespace {
std::atomic<long long> g_launches{ 0 }, g_allocs{ 0 };
struct NullFunction { int rolled; };
struct NullModule { int rolled; NullFunction fn; };
struct NullProgram { std::string source; std::string code; };
void* aligned(size_t bytes) { void* p = nullptr; if (posix_memalign(&p, 256, bytes ? bytes : 256) != 0) return nullptr; return p; }

void touch_rows(const fm::DevProgramArgs& a, const uint64_t* rows, uint32_t batch) {
    const uint64_t* table = a.use_inline ? a.inline_row : rows;
    if (!table || a.n <= 0) return;
    for (uint32_t b = 0; b < batch; ++b) {
        const uint64_t* r = table + (size_t)b * a.row_words;
        for (uint32_t k = 0; k < a.n_in; ++k) { const volatile float* p = reinterpret_cast<const float*>(r[k]); (void)p[0]; (void)p[a.n - 1]; }
        for (uint32_t k = 0; k < a.n_out; ++k) { volatile float* p = reinterpret_cast<float*>(r[a.n_in + k]); p[0] = 0.5f; p[a.n - 1] = 0.25f; }
    }
}
void write_moments(double* results, size_t count, int64_t n, uint64_t* done_flag, uint64_t done_value) {
    if (results) for (size_t i = 0; i < count; ++i) { results[4 * i] = 0.5 * (double)n; results[4 * i + 1] = 0.3 * (double)n; results[4 * i + 2] = 0.0; results[4 * i + 3] = 1.0; }
    if (done_flag) __atomic_store_n(done_flag, done_value, __ATOMIC_RELEASE);
}
}

extern "C" long long fmnull_launches() { return g_launches.load(); }

namespace fm {
hipError_t launch_program(const DevProgramArgs& a, const uint64_t* rows, double*, uint32_t, uint32_t batch, hipStream_t) {
    ++g_launches;
    touch_rows(a, rows, batch);
    if (a.n_red > 0) write_moments(a.results, (size_t)batch * a.n_red, a.n, a.done_flag, a.done_value);
    return hipSuccess;
}
hipError_t launch_bm(const DevBmArgs& a, uint32_t n_streams, hipStream_t) {
    ++g_launches;
    for (uint32_t s = 0; s < n_streams; ++s) { volatile float* p = a.slab + (size_t)s * a.stride_floats; if (a.n_paths > 0) { p[0] = 0.01f; p[a.n_paths - 1] = -0.01f; } }
    return hipSuccess;
}
hipError_t launch_gather_moments(const DevGatherArgs& a, double* out, hipStream_t) {
    ++g_launches;
    for (uint32_t i = 0; i < a.count; ++i) std::memcpy(out + (size_t)i * 4, reinterpret_cast<const void*>((uintptr_t)a.src[i]), 32);
    return hipSuccess;
}
hipError_t preload_kernels() { return hipSuccess; }
hipError_t launch_combine_moments(const double* gathered, uint32_t world, uint32_t count, double* out, hipStream_t) {       // (computed: the rule is small enough to state twice)
    ++g_launches;
    for (uint32_t k = 0; k < count; ++k) {
        double m[4]; std::memcpy(m, gathered + (size_t)k * 4, 32);
        for (uint32_t r = 1; r < world; ++r) {
            const double* g = gathered + ((size_t)r * count + k) * 4;
            m[0] += g[0]; m[1] += g[1];
            m[2] = (m[2] != m[2] || g[2] != g[2]) ? __builtin_nan("") : (g[2] < m[2] || (g[2] == m[2] && __builtin_signbit(g[2]))) ? g[2] : m[2];
            m[3] = (m[3] != m[3] || g[3] != g[3]) ? __builtin_nan("") : (g[3] > m[3] || (g[3] == m[3] && !__builtin_signbit(g[3]))) ? g[3] : m[3];
        }
        std::memcpy(out + (size_t)k * 4, m, 32);
    }
    return hipSuccess;
}
hipError_t launch_fill(float* p, float v, int64_t n_padded, hipStream_t) { ++g_launches; for (int64_t i = 0; i < n_padded; ++i) p[i] = v; return hipSuccess; }
}

extern "C" {
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "null-device error"; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_t* p, int) { std::memset(p, 0, sizeof *p); std::strcpy(p->name, "null device (tests)"); std::strcpy(p->gcnArchName, "gfx950"); p->multiProcessorCount = 256; p->totalGlobalMem = size_t(288) << 30; return hipSuccess; }
// FMNULL_DEVICE_BYTES=N: a device of N bytes — allocations beyond it fail, hipMemGetInfo tells what is left (the engine's behaviour when
// device memory runs out: purge, FMHIP_ERR_OUT_OF_MEMORY, the caller's collection, the retry)
static const size_t DEVICE_BYTES = [] { const char* e = std::getenv("FMNULL_DEVICE_BYTES"); return e ? (size_t)std::atoll(e) : (size_t(288) << 30); }();
static std::mutex g_mem_mu;
static std::unordered_map<void*, size_t> g_mem;
static size_t g_mem_used = 0;
hipError_t hipMemGetInfo(size_t* fr, size_t* tot) { std::lock_guard<std::mutex> lock(g_mem_mu); *fr = DEVICE_BYTES - std::min(DEVICE_BYTES, g_mem_used); *tot = DEVICE_BYTES; return hipSuccess; }
hipError_t hipMalloc(void** p, size_t bytes) {
    ++g_allocs;
    std::lock_guard<std::mutex> lock(g_mem_mu);
    if (g_mem_used + bytes > DEVICE_BYTES) { *p = nullptr; return hipErrorOutOfMemory; }
    *p = aligned(bytes);
    if (!*p) return hipErrorOutOfMemory;
    g_mem[*p] = bytes; g_mem_used += bytes;
    return hipSuccess;
}
hipError_t hipFree(void* p) { { std::lock_guard<std::mutex> lock(g_mem_mu); auto it = g_mem.find(p); if (it != g_mem.end()) { g_mem_used -= it->second; g_mem.erase(it); } } std::free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) { *p = aligned(bytes); if (*p) std::memset(*p, 0, bytes); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { if (n) std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { if (n) std::memset(d, v, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { if (n) std::memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = reinterpret_cast<hipStream_t>(new int(0)); return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t* s) { return hipStreamCreateWithFlags(s, 0); }
hipError_t hipStreamDestroy(hipStream_t s) { delete reinterpret_cast<int*>(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = reinterpret_cast<hipEvent_t>(new int(0)); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete reinterpret_cast<int*>(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { *reinterpret_cast<volatile int*>(e) = 1; return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return hipSuccess; }
hipError_t hipModuleLoadData(hipModule_t* m, const void* image) {
    if (std::memcmp(image, "FMNULLCO", 8) != 0) return hipErrorInvalidImage;
    { const int rolled = ((const char*)image)[8] == 'R' ? 1 : 0; *m = reinterpret_cast<hipModule_t>(new NullModule{ rolled, NullFunction{ rolled } }); }
    return hipSuccess;
}
hipError_t hipModuleUnload(hipModule_t m) { delete reinterpret_cast<NullModule*>(m); return hipSuccess; }
hipError_t hipModuleGetFunction(hipFunction_t* f, hipModule_t m, const char*) { *f = reinterpret_cast<hipFunction_t>(&reinterpret_cast<NullModule*>(m)->fn); return hipSuccess; }
hipError_t hipFuncGetAttribute(int* v, hipFunction_attribute, hipFunction_t) { *v = 64; return hipSuccess; }
hipError_t hipModuleLaunchKernel(hipFunction_t f, unsigned, unsigned gy, unsigned, unsigned, unsigned, unsigned, unsigned, hipStream_t, void** params, void**) {
    ++g_launches;
    if (reinterpret_cast<NullFunction*>(f)->rolled) {
        const fm::DevRolledArgs& a = *reinterpret_cast<const fm::DevRolledArgs*>(params[0]);
        if (a.results) write_moments(a.results, (size_t)gy * (a.pad ? a.pad : 1u), a.n, a.done_flag, a.done_value);      // (pad: chains per row of a merged launch)
    } else {
        const fm::DevProgramArgs& a = *reinterpret_cast<const fm::DevProgramArgs*>(params[0]);
        touch_rows(a, *reinterpret_cast<const uint64_t* const*>(params[1]), gy);
        if (a.n_red > 0) write_moments(a.results, (size_t)gy * a.n_red, a.n, a.done_flag, a.done_value);
    }
    return hipSuccess;
}

hiprtcResult hiprtcVersion(int* major, int* minor) { *major = 0; *minor = 0; return HIPRTC_SUCCESS; }
const char* hiprtcGetErrorString(hiprtcResult) { return "null-hiprtc error"; }
hiprtcResult hiprtcCreateProgram(hiprtcProgram* prog, const char* src, const char*, int, const char* const*, const char* const*) {
    NullProgram* p = new NullProgram{ src, "" };
    *prog = reinterpret_cast<hiprtcProgram>(p);
    return HIPRTC_SUCCESS;
}
hiprtcResult hiprtcCompileProgram(hiprtcProgram prog, int, const char* const*) {
    NullProgram* p = reinterpret_cast<NullProgram*>(prog);
    // FMNULL_COMPILE_MS: a compilation takes that long (the real hiprtc needs seconds per kernel: callers that meet a kernel another
    // thread is compiling have to wait for it, and be woken)
    if (const char* e = std::getenv("FMNULL_COMPILE_MS")) { timespec ts{ 0, 0 }; const long ms = std::atol(e); ts.tv_sec = ms / 1000; ts.tv_nsec = (ms % 1000) * 1000000L; nanosleep(&ts, nullptr); }
    p->code = std::string("FMNULLCO") + (p->source.find("DevRolledArgs") != std::string::npos ? "R" : "P") + std::string(7, '\0');
    return HIPRTC_SUCCESS;
}
hiprtcResult hiprtcGetProgramLogSize(hiprtcProgram, size_t* n) { *n = 0; return HIPRTC_SUCCESS; }
hiprtcResult hiprtcGetProgramLog(hiprtcProgram, char*) { return HIPRTC_SUCCESS; }
hiprtcResult hiprtcGetCodeSize(hiprtcProgram prog, size_t* n) { *n = reinterpret_cast<NullProgram*>(prog)->code.size(); return HIPRTC_SUCCESS; }
hiprtcResult hiprtcGetCode(hiprtcProgram prog, char* out) { const std::string& c = reinterpret_cast<NullProgram*>(prog)->code; std::memcpy(out, c.data(), c.size()); return HIPRTC_SUCCESS; }
hiprtcResult hiprtcDestroyProgram(hiprtcProgram* prog) { delete reinterpret_cast<NullProgram*>(*prog); *prog = nullptr; return HIPRTC_SUCCESS; }
}
