// fma_pow2_exhaustive.hip — is fma(x, s, 1) the same float as (x * s) + 1 (two roundings, -ffp-contract=off) when s is a power of two?
// By argument: x * s is exact for s = ±2^k unless it overflows (both forms give ±inf) or underflows (both forms give exactly 1: what is
// lost is below 2^-126, half an ulp of 1 is 2^-25).  By enumeration, here: every normal power of two s (2 x 254) against all 2^32 x.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math benchmarks/fma_pow2_exhaustive.hip -o benchmarks/build/fma_pow2_exhaustive
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Out { unsigned long long differ; uint32_t first_x, first_s; };

// thread = 2^12 consecutive x; blockIdx.y = the scalar
__global__ void __launch_bounds__(256) sweep(Out* out) {
    const uint32_t e = blockIdx.y >> 1, sign = blockIdx.y & 1u;
    const float s = __uint_as_float((sign << 31) | ((e + 1u) << 23));            // biased exponents 1 … 254
    const uint32_t base = (blockIdx.x * 256u + threadIdx.x) << 12;
    unsigned bad = 0; uint32_t fx = 0;
    for (uint32_t i = 0; i < 4096u; ++i) {
        const float x = __uint_as_float(base + i);
        const float p = x * s;
        const float two = 1.0f + p;
        const float one = __builtin_fmaf(x, s, 1.0f);
        const uint32_t a = __float_as_uint(two), b = __float_as_uint(one);
        const bool nan_a = (a & 0x7fffffffu) > 0x7f800000u, nan_b = (b & 0x7fffffffu) > 0x7f800000u;
        if (nan_a || nan_b ? (nan_a != nan_b || a != b) : a != b) { if (!bad) fx = base + i; ++bad; }
    }
    if (bad) { if (atomicAdd(&out->differ, (unsigned long long)bad) == 0ull) { out->first_x = fx; out->first_s = __float_as_uint(s); } }
}

int main() {
    Out* dev = nullptr; CK(hipMalloc(&dev, sizeof(Out))); CK(hipMemset(dev, 0, sizeof(Out)));
    sweep<<<dim3((1u << 20) / 256u, 508), 256>>>(dev);
    CK(hipDeviceSynchronize());
    Out h{}; CK(hipMemcpy(&h, dev, sizeof h, hipMemcpyDeviceToHost));
    printf("{\"what\": \"fma(x, s, 1) against (x * s) + 1 for every normal power of two s (508) and all 2^32 x, NaN results compared bit for bit too\", \"evaluations\": %.0f, \"differ\": %llu, \"first_x_bits\": \"0x%08x\", \"first_s_bits\": \"0x%08x\"}\n",
           508.0 * 4294967296.0, h.differ, h.first_x, h.first_s);
    return 0;
}
