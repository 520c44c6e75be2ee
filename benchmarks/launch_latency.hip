// launch_latency.hip — what "an empty launch between two events" (8.4 µs on MI355X) is made of.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 benchmarks/launch_latency.hip -o benchmarks/build/launch_latency
// Legs (median of 2000 repetitions each, stream 0 idle before every repetition):
//   events_only         e0, e1 recorded back to back: the cost of the two event packets themselves
//   empty_between       e0, empty<<<1,64>>>, e1: the figure quoted so far
//   empty_own_events    the same kernel with the events INSIDE its dispatch packet (hipExtLaunchKernelGGL start/stop): begin … end of the
//                       kernel as the command processor stamps it — no barrier packets around it
//   row_between / row_own   a one-row kernel of the size of stream S on 10^6 paths (3 reads, 1 write of 4 MB each, no arithmetic worth naming)
//   back_to_back        (e0, 64 empty launches, e1) / 64: what a launch costs when the queue is never empty
//   host_enqueue        host time of one hipLaunchKernelGGL call (queue never waits)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void empty_kernel() {}
__global__ void __launch_bounds__(256) row_kernel(const float4* a, const float4* b, const float4* c, float4* o, long n4) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += gridDim.x * 256L) {
        float4 x = a[i], y = b[i], z = c[i];
        o[i] = make_float4(x.x + y.x * z.x, x.y + y.y * z.y, x.z + y.z * z.z, x.w + y.w * z.w);
    }
}
static double median(std::vector<float>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
    const int REPS = 2000;
    const long n = 1000000, n4 = n / 4;
    float4 *a, *b, *c, *o;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&c, n * 4)); CK(hipMalloc(&o, n * 4));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4)); CK(hipMemset(c, 0, n * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int row_grid = (int)((n4 + 255) / 256 / 2);
    std::vector<float> t((size_t)REPS);
    auto leg = [&](const char* name, auto&& body) -> int {
        for (int r = 0; r < REPS + 50; ++r) {
            CK(hipDeviceSynchronize());
            body();
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 50) t[(size_t)(r - 50)] = ms * 1e3f;
        }
        std::vector<float> s = t; const double med = median(s);
        printf("  \"%s\": {\"median_us\": %.2f, \"p10_us\": %.2f, \"p90_us\": %.2f},\n", name, med, s[s.size() / 10], s[s.size() * 9 / 10]);
        return 0;
    };
    printf("{\n");
    if (leg("events_only", [&] { (void)hipEventRecord(e0, 0); (void)hipEventRecord(e1, 0); })) return 1;
    if (leg("empty_between", [&] { (void)hipEventRecord(e0, 0); hipLaunchKernelGGL(empty_kernel, 1, 64, 0, 0); (void)hipEventRecord(e1, 0); })) return 1;
    if (leg("empty_own_events", [&] { hipExtLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0, e0, e1, 0); })) return 1;
    if (leg("row_between", [&] { (void)hipEventRecord(e0, 0); hipLaunchKernelGGL(row_kernel, row_grid, 256, 0, 0, a, b, c, o, n4); (void)hipEventRecord(e1, 0); })) return 1;
    if (leg("row_own_events", [&] { hipExtLaunchKernelGGL(row_kernel, dim3(row_grid), dim3(256), 0, 0, e0, e1, 0, a, b, c, o, n4); })) return 1;
    {
        std::vector<float> s;
        for (int r = 0; r < 200; ++r) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            for (int k = 0; k < 64; ++k) hipLaunchKernelGGL(empty_kernel, 1, 64, 0, 0);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); s.push_back(ms * 1e3f / 64);
        }
        printf("  \"back_to_back_empty\": {\"median_us_per_launch\": %.2f},\n", median(s));
        s.clear();
        for (int r = 0; r < 200; ++r) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            for (int k = 0; k < 64; ++k) hipLaunchKernelGGL(row_kernel, row_grid, 256, 0, 0, a, b, c, o, n4);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); s.push_back(ms * 1e3f / 64);
        }
        printf("  \"back_to_back_row\": {\"median_us_per_launch\": %.2f, \"GBps\": %.0f},\n", median(s), 16.0 * n / (median(s) * 1e-6) / 1e9);
    }
    {
        CK(hipDeviceSynchronize());
        const auto h0 = std::chrono::steady_clock::now();
        for (int k = 0; k < 2000; ++k) hipLaunchKernelGGL(empty_kernel, 1, 64, 0, 0);
        const auto h1 = std::chrono::steady_clock::now();
        CK(hipDeviceSynchronize());
        printf("  \"host_enqueue\": {\"us_per_call\": %.2f},\n", std::chrono::duration<double, std::micro>(h1 - h0).count() / 2000);
    }
    {   // the plain kernel at the headline's size: 64 x 10^6 elements, three vectors read, one written (1.024 GB)
        const long N = 64000000, N4 = N / 4;
        float4 *A, *B, *C, *O;
        CK(hipMalloc(&A, N * 4)); CK(hipMalloc(&B, N * 4)); CK(hipMalloc(&C, N * 4)); CK(hipMalloc(&O, N * 4));
        CK(hipMemset(A, 0x3c, N * 4)); CK(hipMemset(B, 0x3c, N * 4)); CK(hipMemset(C, 0x3c, N * 4));
        for (int grid : { 7872, 31250, 62500 }) {
            std::vector<float> s;
            for (int r = 0; r < 6; ++r) {
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, 0));
                for (int k = 0; k < 50; ++k) hipLaunchKernelGGL(row_kernel, grid, 256, 0, 0, A, B, C, O, N4);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r > 0) s.push_back(ms * 1e3f / 50);
            }
            const double us = median(s);
            printf("  \"plain_triad_64M_grid_%d\": {\"us\": %.1f, \"GBps\": %.0f, \"frac\": %.4f},\n", grid, us, 16.0 * N / (us * 1e-6) / 1e9, 16.0 * N / (us * 1e-6) / 8e12);
        }
    }
    printf("  \"what\": \"HIP events on the null stream; medians of %d repetitions, device idle before each\"\n}\n", REPS);
    return 0;
}
