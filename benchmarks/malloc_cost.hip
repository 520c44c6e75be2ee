// Cost of device allocations by size (the pool's slabs: runtime.cpp, Pool::alloc): hipMalloc / first touch / hipFree.
//   hipcc --offload-arch=gfx950 -O2 -o malloc_cost malloc_cost.hip && ./malloc_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
int main() {
    using clk = std::chrono::steady_clock;
    hipFree(nullptr);
    for (size_t mb : { 4, 16, 64, 256, 1024, 4096 }) {
        const size_t bytes = mb << 20;
        const int reps = mb >= 1024 ? 4 : 16;
        std::vector<void*> p((size_t)reps);
        auto t0 = clk::now();
        for (int i = 0; i < reps; ++i) if (hipMalloc(&p[(size_t)i], bytes) != hipSuccess) { std::printf("hipMalloc failed\n"); return 1; }
        auto t1 = clk::now();
        for (int i = 0; i < reps; ++i) (void)hipMemsetAsync(p[(size_t)i], 0, bytes, nullptr);
        (void)hipDeviceSynchronize();
        auto t2 = clk::now();
        for (int i = 0; i < reps; ++i) (void)hipMemsetAsync(p[(size_t)i], 0, bytes, nullptr);
        (void)hipDeviceSynchronize();
        auto t3 = clk::now();
        for (int i = 0; i < reps; ++i) (void)hipFree(p[(size_t)i]);
        auto t4 = clk::now();
        auto us = [](clk::duration d) { return std::chrono::duration<double, std::micro>(d).count(); };
        std::printf("%5zu MB: hipMalloc %9.1f us (%.2f us/MB)   first memset %9.1f us   second memset %9.1f us   hipFree %9.1f us\n", mb, us(t1 - t0) / reps, us(t1 - t0) / reps / mb,
                    us(t2 - t1) / reps, us(t3 - t2) / reps, us(t4 - t3) / reps);
    }
    return 0;
}
