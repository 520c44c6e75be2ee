// launch_cost.hip — what a kernel launch costs the calling thread, by kernel-argument size and by what the stream was doing:
// back to back on a busy stream, after a hipStreamSynchronize, after polling a flag in pinned memory instead.
//   hipcc --offload-arch=gfx950 -O3 benchmarks/launch_cost.hip -o benchmarks/build/launch_cost && benchmarks/build/launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
struct Small { float* p; uint64_t* flag; uint64_t value; int n; };
struct Big { float* p; uint64_t* flag; uint64_t value; int n; uint64_t pad[440]; };          // ≈ 3.5 KB, like fm::DevProgramArgs
template <class A> __global__ void k(const A a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < a.n) a.p[i] = a.p[i] * 1.0001f + 1.0f;
    if (a.flag && i == 0) __hip_atomic_store(a.flag, a.value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
using clk = std::chrono::steady_clock;
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }

template <class A> int run(const char* name, hipStream_t st, float* buf, uint64_t* flag, int n, int work_blocks) {
    A a{}; a.p = buf; a.n = n;
    const int N = 2000;
    for (int mode = 0; mode < 4; ++mode) {                 // 0: back to back; 1: sync after every launch; 2: poll after every launch; 3: three launches, then sync
        double api = 0.0, total = 0.0; uint64_t seq = 0;
        CK(hipStreamSynchronize(st));
        const auto T0 = clk::now();
        for (int i = 0; i < N; ++i) {
            a.flag = mode == 2 ? flag : nullptr; a.value = ++seq;
            const auto t0 = clk::now();
            k<A><<<work_blocks, 256, 0, st>>>(a);
            api += us(t0, clk::now());
            if (mode == 1 || (mode == 3 && i % 3 == 2)) CK(hipStreamSynchronize(st));
            if (mode == 2) { while (*(volatile uint64_t*)flag != seq) { } }
        }
        CK(hipStreamSynchronize(st));
        total = us(T0, clk::now());
        static const char* modes[4] = { "back to back", "sync after each", "poll after each", "3 launches, sync" };
        printf("%-28s %-18s launch call %6.2f us, loop %7.2f us per launch\n", name, modes[mode], api / N, total / N);
    }
    return 0;
}
int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int n = 1 << 20;
    float* buf; CK(hipMalloc(&buf, (size_t)n * 4)); CK(hipMemset(buf, 0, (size_t)n * 4));
    uint64_t* flag; CK(hipHostMalloc((void**)&flag, 64, hipHostMallocDefault)); *flag = 0;
    if (run<Small>("32-byte arguments, 4 MB", st, buf, flag, n, n / 256)) return 1;
    if (run<Big>("3.5 KB arguments, 4 MB", st, buf, flag, n, n / 256)) return 1;
    if (run<Big>("3.5 KB arguments, tiny", st, buf, flag, 256, 1)) return 1;
    return 0;
}
