// VALU cost of the per-element bodies of the fused kernels (fm_device_math.hpp), measured WITHOUT memory traffic: every lane
// iterates a body over 8 register-resident elements, 4 waves per SIMD (the occupancy of the specialised kernels).  Prints ns per
// element and SIMD for each body — to be read against the memory time of the bench launch per element and SIMD
// (64 M elements in ≈ 160 µs on 1024 SIMDs = 2.56 ns) — and checks candidate bodies against the shipped ones over ALL 2^32
// fp32 bit patterns (the shipped exp / log / sqrt are bit-identical to the CPU twin over all 2^32 inputs, so "identical to
// shipped" = "identical to the twin").
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I finmath-lib-cuda-extensions_amd/csrc benchmarks/valu_cost.hip -o /tmp/valu && /tmp/valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "fm_device_math.hpp"
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
using namespace fm;

// ---- candidates -------------------------------------------------------------------------------------------------------
// sqrt: hardware reciprocal square root + Newton steps on FMA residuals (Markstein): y0 = x·r, h = r/2, e = x - y0² (exact in
// one FMA), y1 = y0 + e·h; a second step makes a faithful y1 correctly rounded.  Zero, infinity, NaN, negative and denormal
// arguments (rsq gives inf / NaN / loses bits there) take the IEEE expansion, behind a wave-uniform branch.
__device__ __forceinline__ float sqrt_newton1(float x) {
    const float r = __builtin_amdgcn_rsqf(x);
    const float y0 = x * r, h = 0.5f * r;
    const float e = __builtin_fmaf(-y0, y0, x);
    return __builtin_fmaf(e, h, y0);
}
__device__ __forceinline__ float sqrt_newton2(float x) {
    const float r = __builtin_amdgcn_rsqf(x);
    const float y0 = x * r, h = 0.5f * r;
    const float e0 = __builtin_fmaf(-y0, y0, x);
    const float y1 = __builtin_fmaf(e0, h, y0);
    const float e1 = __builtin_fmaf(-y1, y1, x);
    return __builtin_fmaf(e1, h, y1);
}
__device__ __forceinline__ float sqrt_hw_newton(float x) {          // v_sqrt_f32 as the start, rsq for the slope
    const float y0 = __builtin_amdgcn_sqrtf(x), h = 0.5f * __builtin_amdgcn_rsqf(x);
    const float e = __builtin_fmaf(-y0, y0, x);
    return __builtin_fmaf(e, h, y0);
}
template <int V> __device__ __forceinline__ float sqrt_variant(float x) {
    // +normal only (class 0x100); everything else: the IEEE expansion
    const bool plain = (__float_as_uint(x) - 0x20000000u) < 0x3F000000u;        // x in [2^-63, 2^63): positive, normal, residuals stay normal
    float fast;
    if constexpr (V == 1) fast = sqrt_newton1(x); else if constexpr (V == 2) fast = sqrt_newton2(x); else fast = sqrt_hw_newton(x);
    if (__builtin_amdgcn_ballot_w64(!plain) != 0ull) {          // wave-uniform, rare; the volatile asm keeps the compiler from if-converting it
        asm volatile("; sqrt: IEEE expansion for special arguments");
        const float slow = __builtin_sqrtf(x);
        return plain ? fast : slow;
    }
    return fast;
}

struct BodyBase   { static __device__ __forceinline__ float f(float a) { return a * 0.99f + 0.01f; } };
struct BodyExp    { static __device__ __forceinline__ float f(float a) { return exp_f(a) * 0.25f; } };
struct BodyLog    { static __device__ __forceinline__ float f(float a) { return log_f<true>(a) + 2.0f; } };
struct BodySqrt   { static __device__ __forceinline__ float f(float a) { return sqrt_f(a) + 1.0f; } };
struct BodySqrt1  { static __device__ __forceinline__ float f(float a) { return sqrt_variant<1>(a) + 1.0f; } };
struct BodySqrt2  { static __device__ __forceinline__ float f(float a) { return sqrt_variant<2>(a) + 1.0f; } };
struct BodySqrt3  { static __device__ __forceinline__ float f(float a) { return sqrt_variant<3>(a) + 1.0f; } };
struct BodyExpFast{ static __device__ __forceinline__ float f(float a) { return exp_fast(a) * 0.25f; } };
struct BodyLogFast{ static __device__ __forceinline__ float f(float a) { return log_fast(a) + 2.0f; } };
struct BodyDiv    { static __device__ __forceinline__ float f(float a) { return 1.0f / a + 0.5f; } };
struct BodyRed    { static __device__ __forceinline__ float f(float a) { return a; } };      // the reduction accumulation alone (see kernel)

template <typename B, bool RED>
__global__ void __launch_bounds__(256) body_kernel(const float* __restrict__ in, float* __restrict__ out, int iters) {
    __shared__ int dummy;
    if (threadIdx.x == 0) dummy = 0;
    log_table_init();
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = in[(blockIdx.x * 256 + threadIdx.x) * 8 + j];
    double s1 = 0.0, s2 = 0.0; float mn = 1e30f, mx = -1e30f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; g += 4) {
#pragma unroll
            for (int j = g; j < g + 4; ++j) a[j] = B::f(a[j]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (RED) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const double d = (double)a[j]; s1 += d; s2 = __builtin_fma(d, d, s2); }
#pragma unroll
            for (int j = 0; j < 8; j += 2) { mn = hw_min3(mn, a[j], a[j + 1]); mx = hw_max3(mx, a[j], a[j + 1]); }
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = a[j] * 0.99f + 0.01f;
        }
    }
    float r = (float)(s1 + s2) + mn + mx + (float)dummy;
#pragma unroll
    for (int j = 0; j < 8; ++j) r += a[j];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}


// ---- normal-increment generator: shipped (kernels.hip, IEEE-only Box-Muller) against the CDNA4 re-specification -------------
__device__ __forceinline__ void philox10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float old_logf(float u) {
    const uint32_t b = __float_as_uint(u);
    int e = (int)(b >> 23) - 127;
    float f = __uint_as_float((b & 0x007FFFFFu) | 0x3F800000u);
    if (f > 1.41421354f) { f = f * 0.5f; e += 1; }
    const float s = (f - 1.0f) / (f + 1.0f), z = s * s;
    float p = 0.222222224f; p = __builtin_fmaf(p, z, 0.285714298f); p = __builtin_fmaf(p, z, 0.400000006f); p = __builtin_fmaf(p, z, 0.666666687f); p = p * z;
    const float lnf = __builtin_fmaf(s, p, s + s), ef = (float)e;
    return __builtin_fmaf(ef, 0.693145751953125f, __builtin_fmaf(ef, 1.42860677e-06f, lnf));
}
__device__ __forceinline__ void old_box_muller(uint32_t ra, uint32_t rb, float& za, float& zb) {
    const float u1 = __builtin_fmaf((float)ra, 0x1p-32f, 0x1p-33f);
    const float radius = __builtin_sqrtf(-2.0f * old_logf(u1));
    const float t = (float)(rb >> 8) * 0x1p-22f; const int q = (int)t; const float fr = t - (float)q;
    const bool swap = fr > 0.5f; const float g = swap ? 1.0f - fr : fr; const float x = g * 1.57079637f, x2 = x * x;
    float ps = 2.75573188e-06f; ps = __builtin_fmaf(ps, x2, -1.98412701e-04f); ps = __builtin_fmaf(ps, x2, 8.33333377e-03f); ps = __builtin_fmaf(ps, x2, -1.66666672e-01f); ps = ps * x2;
    const float sinx = __builtin_fmaf(x, ps, x);
    float pc = -2.75573192e-07f; pc = __builtin_fmaf(pc, x2, 2.48015876e-05f); pc = __builtin_fmaf(pc, x2, -1.38888892e-03f); pc = __builtin_fmaf(pc, x2, 4.16666679e-02f); pc = __builtin_fmaf(pc, x2, -0.5f);
    const float cosx = __builtin_fmaf(pc, x2, 1.0f);
    const float sp = swap ? cosx : sinx, cp = swap ? sinx : cosx;
    float c, sn;
    switch (q & 3) { case 0: c = cp; sn = sp; break; case 1: c = -sp; sn = cp; break; case 2: c = -cp; sn = -sp; break; default: c = sp; sn = -cp; break; }
    za = radius * c; zb = radius * sn;
}
// re-specification: no division, no IEEE-expanded sqrt, integer range reduction, D4 symmetry from three random bits
__device__ __forceinline__ void new_box_muller(uint32_t ra, uint32_t rb, float& za, float& zb) {
    const uint32_t k = ra | 1u;
    const uint32_t lz = (uint32_t)__builtin_clz(k);
    const float M = (float)(k << lz);                                   // [2^31, 2^32], RNE
    const float t = __builtin_fmaf(M, 0x1p-31f, -1.5f);
    float q = 0x1.89a61cp-7f;
    q = __builtin_fmaf(q, t, -0x1.4f7c14p-6f); q = __builtin_fmaf(q, t, 0x1.d8347ap-6f); q = __builtin_fmaf(q, t, -0x1.a97d34p-5f);
    q = __builtin_fmaf(q, t, 0x1.94abecp-4f);  q = __builtin_fmaf(q, t, -0x1.94a656p-3f); q = __builtin_fmaf(q, t, 0x1.c71c4ap-2f);
    q = __builtin_fmaf(q, t, -0x1.555544p+0f); q = __builtin_fmaf(q, t, 0x1.269622p-1f);
    float r2 = __builtin_fmaf((float)lz, 0x1.62e43p+0f, q);             // lz · 2 ln 2 + Q(t)
    r2 = __builtin_fmaxf(r2, 0x1p-40f);
    const float rs = __builtin_amdgcn_rsqf(r2);                          // sqrt: rsq + one Newton step on the exact FMA residual
    const float y0 = r2 * rs, h = 0.5f * rs;
    const float radius = __builtin_fmaf(__builtin_fmaf(-y0, y0, r2), h, y0);
    const float x = (float)(rb & 0x1fffffffu) * 0x1.921fb6p-30f;        // [0, π/4): 29 bits · π/4 · 2^-29
    const float x2 = x * x;
    float ps = -0x1.9ac9bp-13f; ps = __builtin_fmaf(ps, x2, 0x1.110c28p-7f); ps = __builtin_fmaf(ps, x2, -0x1.555552p-3f); ps = ps * x2;
    const float sinx = __builtin_fmaf(x, ps, x);
    float pc = -0x1.6c0e08p-10f; pc = __builtin_fmaf(pc, x2, 0x1.55554cp-5f); pc = __builtin_fmaf(pc, x2, -0.5f);
    const float cosx = __builtin_fmaf(pc, x2, 1.0f);
    const uint32_t swap = (uint32_t)((int32_t)(rb << 2) >> 31);          // bit 29 -> all ones / zero
    const uint32_t cb = __float_as_uint(cosx), sb = __float_as_uint(sinx);
    const float a = __uint_as_float((sb & swap) | (cb & ~swap)), b = __uint_as_float((cb & swap) | (sb & ~swap));
    const float pa = radius * a, pb2 = radius * b;                       // both >= 0
    za = __uint_as_float((__float_as_uint(pa) & 0x7fffffffu) | (rb & 0x80000000u));           // sign from bit 31
    zb = __uint_as_float((__float_as_uint(pb2) & 0x7fffffffu) | ((rb << 1) & 0x80000000u));   // sign from bit 30
}
template <int MODE>      // 0: shipped Philox + shipped Box-Muller, 1: Philox only, 2: new Box-Muller only, 3: Philox + new Box-Muller
__global__ void __launch_bounds__(256) normal_kernel(float* __restrict__ out, int iters, float sq) {
    const uint32_t lane = blockIdx.x * 256 + threadIdx.x;
    float acc = 0.0f; uint32_t xacc = 0;
    uint32_t ra = lane * 2654435761u, rb = lane * 40503u + 77u;
    for (int it = 0; it < iters; ++it) {
        uint32_t r[4];
        if constexpr (MODE == 2) { ra += 0x9E3779B9u; rb += 0x7F4A7C15u; r[0] = ra; r[1] = rb; r[2] = ra ^ 0x5bd1e995u; r[3] = rb ^ 0x27d4eb2fu; }
        else philox10(lane, (uint32_t)it, 7u, 0x464D4850u, 31415u, 0u, r);
        if constexpr (MODE == 1) { xacc ^= r[0] ^ r[1] ^ r[2] ^ r[3]; }
        else {
            float z[4];
            if constexpr (MODE == 0) { old_box_muller(r[0], r[1], z[0], z[1]); old_box_muller(r[2], r[3], z[2], z[3]); }
            else { new_box_muller(r[0], r[1], z[0], z[1]); new_box_muller(r[2], r[3], z[2], z[3]); }
            acc += sq * z[0]; acc += sq * z[1]; acc += sq * z[2]; acc += sq * z[3];
        }
    }
    out[lane] = acc + (float)xacc;
}
template <int MODE>
static int time_normal(const char* name, float* out) {
    const int iters = 2000, blocks = 1024;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    normal_kernel<MODE><<<blocks, 256>>>(out, 10, 0.7f);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0)); normal_kernel<MODE><<<blocks, 256>>>(out, iters, 0.7f); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double normals_per_simd = 4.0 * 64 * 4 * iters;
    printf("%-44s %8.3f ms  %7.3f ns per normal and SIMD   (write budget at 6.3 TB/s: 0.650)\n", name, best, best * 1e6 / normals_per_simd);
    return 0;
}
// every r2 the transform can produce is a float in [2^-40, 64): rsq + one Newton step against the IEEE expansion on ALL of them
__global__ void __launch_bounds__(256) check_radius_sqrt(unsigned long long* diff, unsigned int* first) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    unsigned long long local = 0;
    for (uint64_t b = 0x2B800000ull + (uint64_t)blockIdx.x * 256 + threadIdx.x; b < 0x42800000ull; b += stride) {      // [2^-40, 64)
        const float x = __uint_as_float((uint32_t)b);
        const float want = __builtin_sqrtf(x), got = sqrt_newton1(x);
        if (__float_as_uint(want) != __float_as_uint(got)) { ++local; atomicMin(first, (uint32_t)b); }
    }
    if (local) atomicAdd(diff, local);
}

template <typename B, bool RED = false>
static int time_body(const char* name, const float* in, float* out, double base_ns) {
    const int iters = 2000, blocks = 1024;          // 4 workgroups of 4 waves per CU = 4 waves per SIMD
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    body_kernel<B, RED><<<blocks, 256>>>(in, out, 10);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        body_kernel<B, RED><<<blocks, 256>>>(in, out, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double elems_per_simd = 4.0 * 64 * 8 * iters;          // 4 waves x 64 lanes x 8 elements
    const double ns = best * 1e6 / elems_per_simd;
    printf("%-28s %8.3f ms  %7.3f ns per element and SIMD   (minus loop body: %7.3f)\n", name, best, ns, ns - base_ns);
    return 0;
}

// ---- exhaustive comparison of a candidate with the shipped body
template <int V>
__global__ void __launch_bounds__(256) check_sqrt(unsigned long long* diff, unsigned int* first) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    unsigned long long local = 0;
    for (uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x; b < (1ull << 32); b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        const float want = sqrt_f(x), got = sqrt_variant<V>(x);
        const bool same = (__float_as_uint(want) == __float_as_uint(got)) || (want != want && got != got);
        if (!same) { ++local; atomicMin(first, (uint32_t)b); }
    }
    if (local) atomicAdd(diff, local);
}

int main() {
    const int blocks = 1024;
    std::vector<float> h((size_t)blocks * 256 * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.5f + (float)((i * 2654435761u) % 1000) * 1e-3f;
    float *in, *out; CK(hipMalloc(&in, h.size() * 4)); CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    printf("budget: 64 M elements in 160 us on 1024 SIMDs = 2.56 ns per element and SIMD\n");
    time_body<BodyBase>("loop body (mul + add)", in, out, 0.0);
    // second call gives the base to subtract
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0)); body_kernel<BodyBase, false><<<blocks, 256>>>(in, out, 2000); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double base = ms * 1e6 / (4.0 * 64 * 8 * 2000);
    time_body<BodyExp>("exp_f (shipped, fp64)", in, out, base);
    time_body<BodyLog>("log_f (shipped, fp64 table)", in, out, base);
    time_body<BodySqrt>("sqrt_f (shipped, IEEE exp.)", in, out, base);
    time_body<BodySqrt1>("sqrt rsq + 1 Newton", in, out, base);
    time_body<BodySqrt2>("sqrt rsq + 2 Newton", in, out, base);
    time_body<BodySqrt3>("sqrt v_sqrt + rsq + 1 Newton", in, out, base);
    time_body<BodyDiv>("1/x (IEEE division)", in, out, base);
    time_body<BodyExpFast>("exp_fast", in, out, base);
    time_body<BodyLogFast>("log_fast", in, out, base);
    time_body<BodyRed, true>("reduction accumulate (+loop)", in, out, 0.0);

    time_normal<0>("Philox4x32-10 + shipped Box-Muller", out);
    time_normal<1>("Philox4x32-10 alone", out);
    time_normal<2>("re-specified Box-Muller alone", out);
    time_normal<3>("Philox4x32-10 + re-specified Box-Muller", out);
    unsigned long long* diff; unsigned int* first;
    CK(hipMalloc(&diff, 8)); CK(hipMalloc(&first, 4));
    {
        CK(hipMemset(diff, 0, 8)); CK(hipMemset(first, 0xff, 4));
        check_radius_sqrt<<<4096, 256>>>(diff, first); CK(hipDeviceSynchronize());
        unsigned long long d; unsigned int f; CK(hipMemcpy(&d, diff, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost));
        printf("rsq + 1 Newton step vs IEEE sqrt on every float in [2^-40, 64): %llu differences (first at 0x%08x)\n", d, f);
    }
    for (int v = 1; v <= 3; ++v) {
        CK(hipMemset(diff, 0, 8)); CK(hipMemset(first, 0xff, 4));
        if (v == 1) check_sqrt<1><<<4096, 256>>>(diff, first); else if (v == 2) check_sqrt<2><<<4096, 256>>>(diff, first); else check_sqrt<3><<<4096, 256>>>(diff, first);
        CK(hipDeviceSynchronize());
        unsigned long long d; unsigned int f; CK(hipMemcpy(&d, diff, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost));
        printf("sqrt variant %d vs shipped over all 2^32 inputs: %llu differences (first at 0x%08x)\n", v, d, f);
    }
    return 0;
}
