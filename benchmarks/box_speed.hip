// box_speed.hip — what THIS box's memory system and clocks give, in a second: the same binary of the calibration runs at 0.65 of the
// HBM peak on one MI355X box and at 0.71 on another, and a roofline fraction means little without the box's own ceiling
// beside it.  Three kernels over 2 GiB (far beyond the 256 MB memory-side cache), best of 5 launches each, HIP events:
//   copy   out[i] = in[i]            1 read + 1 write, float4, non-temporal          (the guide's "6.29 TB/s measured (float4 copy)")
//   read   a block sum of in[]       read only
//   write  out[i] = v                write only, float4, non-temporal                  (the ceiling of the normal-increment generator)
//   valu   a dependent chain of 4096 packed multiply-adds per lane, no memory        → the clock the chip holds under vector load
// Prints ONE JSON line.   build: csrc/Makefile (bin/box_speed)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("{\"error\": \"%s at line %d\"}\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256) copy_kernel(const f4* __restrict__ in, f4* __restrict__ out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 2048 + threadIdx.x; i < n4 && i < (size_t)(blockIdx.x + 1) * 2048; i += 256) __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}
__global__ void __launch_bounds__(256) write_kernel(f4* __restrict__ out, size_t n4, float v) {
    const f4 x = { v, v, v, v };
    for (size_t i = (size_t)blockIdx.x * 2048 + threadIdx.x; i < n4 && i < (size_t)(blockIdx.x + 1) * 2048; i += 256) __builtin_nontemporal_store(x, out + i);
}
__global__ void __launch_bounds__(256) read_kernel(const f4* __restrict__ in, float* __restrict__ out, size_t n4) {
    f4 acc = { 0.f, 0.f, 0.f, 0.f };
    f4 v[8];
    const size_t base = (size_t)blockIdx.x * 2048 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = base + (size_t)k * 256 < n4 ? __builtin_nontemporal_load(in + base + (size_t)k * 256) : acc;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += v[k];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x] = acc.x;      // (never: keeps the loads)
}
__global__ void __launch_bounds__(256) valu_kernel(float* __restrict__ out, float s) {
    f2 a = { (float)threadIdx.x, 1.0f }, b = { 0.5f, 0.25f }, c = { s, s }, d = { 2.0f, 3.0f };
    for (int i = 0; i < 1024; ++i) {
        a = __builtin_elementwise_fma(a, c, b); d = __builtin_elementwise_fma(d, c, a);
        b = __builtin_elementwise_fma(b, c, d); a = __builtin_elementwise_fma(a, c, b);
        d = __builtin_elementwise_fma(d, c, a); b = __builtin_elementwise_fma(b, c, d);
        a = __builtin_elementwise_fma(a, c, d); d = __builtin_elementwise_fma(d, c, b);
    }
    if (a.x + d.y == 12345.678f) out[threadIdx.x] = a.x + b.y;
}

int main() {
    const size_t bytes = size_t(2) << 30, n4 = bytes / 16;
    f4 *in = nullptr, *out = nullptr;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
    CK(hipMemset(in, 0x3c, bytes)); CK(hipMemset(out, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned blocks = (unsigned)((n4 + 2047) / 2048);
    auto best = [&](auto launch) { float b = 1e30f; for (int r = 0; r < 6; ++r) { (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1); if (r > 0 && ms < b) b = ms; } return b; };
    const float copy_ms = best([&] { copy_kernel<<<blocks, 256>>>(in, out, n4); });
    const float read_ms = best([&] { read_kernel<<<blocks, 256>>>(in, (float*)out, n4); });
    const float write_ms = best([&] { write_kernel<<<blocks, 256>>>(out, n4, 1.5f); });
    // 256 CUs x 4 SIMDs x 8 waves, 8192 dependent packed multiply-adds per lane: 4 cycles each when waves alternate on the SIMD's pipe
    const float valu_ms = best([&] { valu_kernel<<<256 * 8, 256>>>((float*)out, 1.0000001f); });
    CK(hipDeviceSynchronize());
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const double valu_cycles = 8192.0 * 4.0 * 8.0;            // per SIMD: 8 waves x 8192 instructions x 4 cycles
    printf("{\"copy_GBps\": %.0f, \"read_GBps\": %.0f, \"write_GBps\": %.0f, \"valu_clock_GHz\": %.3f, \"bytes\": %zu, \"compute_units\": %d, \"what\": \"float4 non-temporal copy (1 read + 1 write), read-only sweep and write-only fill of 2 GiB, "
           "best of 5 launches; clock = cycles of a dependent packed-FMA chain (8 waves per SIMD, 4 cycles per instruction) / its time\"}\n",
           2.0 * bytes / (copy_ms * 1e-3) / 1e9, (double)bytes / (read_ms * 1e-3) / 1e9, (double)bytes / (write_ms * 1e-3) / 1e9, valu_cycles / (valu_ms * 1e-3) / 1e9, bytes, prop.multiProcessorCount);
    return 0;
}
