// kernarg_size.hip — what a launch costs by the SIZE of its kernel arguments (the engine's kernels take a 3.6 KB argument block by value:
// header + micro-ops + up to 368 inline row words): hipModuleLaunchKernel's host time per launch and launches per second through one stream,
// for argument blocks of 64 B … 3.6 KB.   hipcc --offload-arch=gfx950 -O3 benchmarks/round5/kernarg_size.hip -lhiprtc -o /tmp/kernarg_size
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
using clk = std::chrono::steady_clock;
int main() {
    const int words[] = { 4, 28, 92, 220, 444 };           // + 4 header words of 8 bytes
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float* buf; CK(hipMalloc(&buf, 1 << 22));
    printf("{\"results\": [\n");
    for (size_t v = 0; v < sizeof words / sizeof words[0]; ++v) {
        const int W = words[v];
        std::string src = "struct Args { float* p; unsigned long long n; unsigned long long pad[" + std::to_string(W) + "]; };\n"
                          "extern \"C\" __global__ void k(const Args a, const unsigned long long* rows, double* partials) { const unsigned i = blockIdx.x * 256 + threadIdx.x; if (i < a.n) a.p[i] = (float)a.pad[0]; }\n";
        hiprtcProgram prog; hiprtcCreateProgram(&prog, src.c_str(), "k.hip", 0, nullptr, nullptr);
        const char* opts[] = { "--offload-arch=gfx950", "-O3" };
        if (hiprtcCompileProgram(prog, 2, opts) != HIPRTC_SUCCESS) { printf("compile failed\n"); return 1; }
        size_t size; hiprtcGetCodeSize(prog, &size); std::vector<char> code(size); hiprtcGetCode(prog, code.data());
        hipModule_t mod; CK(hipModuleLoadData(&mod, code.data()));
        hipFunction_t fn; CK(hipModuleGetFunction(&fn, mod, "k"));
        std::vector<uint64_t> args((size_t)W + 2, 0);
        args[0] = (uint64_t)(uintptr_t)buf; args[1] = 256;
        const uint64_t* rows = nullptr; double* partials = nullptr;
        void* params[] = { args.data(), &rows, &partials };
        for (int i = 0; i < 200; ++i) CK(hipModuleLaunchKernel(fn, 1, 1, 1, 256, 1, 1, 0, st, params, nullptr));
        CK(hipStreamSynchronize(st));
        const int K = 20000;
        const auto t0 = clk::now();
        for (int i = 0; i < K; ++i) CK(hipModuleLaunchKernel(fn, 1, 1, 1, 256, 1, 1, 0, st, params, nullptr));
        const auto t1 = clk::now();
        CK(hipStreamSynchronize(st));
        const auto t2 = clk::now();
        printf("%s {\"argument_bytes\": %d, \"host_us_per_launch\": %.3f, \"us_per_launch_through_the_stream\": %.3f}", v ? ",\n" : "", (W + 2) * 8 + 16,
               std::chrono::duration<double, std::micro>(t1 - t0).count() / K, std::chrono::duration<double, std::micro>(t2 - t0).count() / K);
        hipModuleUnload(mod);
    }
    printf("\n]}\n");
    return 0;
}
