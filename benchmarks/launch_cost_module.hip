// launch_cost_module.hip — as launch_cost.hip, for kernels loaded from a code object (hiprtc + hipModuleLaunchKernel, the way the engine
// launches its specialised kernels): several distinct kernels in rotation, three launches and one wait per round.
//   hipcc --offload-arch=gfx950 -O3 benchmarks/launch_cost_module.hip -lhiprtc -o benchmarks/build/launch_cost_module
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
struct Big { float* p; uint64_t* flag; uint64_t value; int n; uint64_t pad[440]; };
using clk = std::chrono::steady_clock;
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }
int main() {
    const int K = 8;
    std::vector<hipFunction_t> fn(K);
    for (int k = 0; k < K; ++k) {
        std::string src = "struct Big { float* p; unsigned long long* flag; unsigned long long value; int n; unsigned long long pad[440]; };\n"
                          "extern \"C\" __global__ void k" + std::to_string(k) + "(const Big a, const unsigned long long* rows, double* partials) { const int i = blockIdx.x * 256 + threadIdx.x; if (i < a.n) a.p[i] = a.p[i] * 1.0001f + " + std::to_string(k + 1) + ".0f; "
                          "if (a.flag && i == 0) __hip_atomic_store(a.flag, a.value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }\n";
        hiprtcProgram prog; hiprtcCreateProgram(&prog, src.c_str(), "k.hip", 0, nullptr, nullptr);
        const char* opts[] = { "--offload-arch=gfx950", "-O3" };
        if (hiprtcCompileProgram(prog, 2, opts) != HIPRTC_SUCCESS) { size_t n; hiprtcGetProgramLogSize(prog, &n); std::string log(n, 0); hiprtcGetProgramLog(prog, &log[0]); printf("%s\n", log.c_str()); return 1; }
        size_t size; hiprtcGetCodeSize(prog, &size); std::vector<char> code(size); hiprtcGetCode(prog, code.data());
        hipModule_t mod; CK(hipModuleLoadData(&mod, code.data()));
        CK(hipModuleGetFunction(&fn[k], mod, ("k" + std::to_string(k)).c_str()));
    }
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int n = 1 << 20;
    float* buf; CK(hipMalloc(&buf, (size_t)n * 4)); CK(hipMemset(buf, 0, (size_t)n * 4));
    uint64_t* flag; CK(hipHostMalloc((void**)&flag, 64, hipHostMallocDefault)); *flag = 0;
    Big a{}; a.p = buf; a.n = n;
    const uint64_t* rows = nullptr; double* partials = nullptr;
    void* params[] = { &a, &rows, &partials };
    const int N = 3000;
    for (int mode = 0; mode < 3; ++mode) {                 // 0: back to back; 1: three launches, then sync; 2: three launches, then poll
        double api = 0.0; uint64_t seq = 0;
        CK(hipStreamSynchronize(st));
        const auto T0 = clk::now();
        for (int i = 0; i < N; ++i) {
            const bool last = i % 3 == 2;
            a.flag = (mode == 2 && last) ? flag : nullptr; a.value = ++seq;
            const auto t0 = clk::now();
            CK(hipModuleLaunchKernel(fn[i % K], n / 256, 1, 1, 256, 1, 1, 0, st, params, nullptr));
            api += us(t0, clk::now());
            if (mode == 1 && last) CK(hipStreamSynchronize(st));
            if (mode == 2 && last) { while (*(volatile uint64_t*)flag != seq) { } }
        }
        CK(hipStreamSynchronize(st));
        static const char* modes[3] = { "back to back", "3 launches, sync", "3 launches, poll" };
        printf("hipModuleLaunchKernel, %d kernels in rotation, 3.5 KB arguments: %-18s launch call %6.2f us, loop %7.2f us per launch\n", K, modes[mode], api / N, us(T0, clk::now()) / N);
    }
    return 0;
}
