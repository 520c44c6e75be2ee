#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

// IEEE-754 correctly rounded a/b for two lanes at once: LLVM's own f32 expansion (AMDGPUISelLowering: LowerFDIV32, denormals on),
// with the six multiply-add steps as packed instructions
__device__ __forceinline__ f2 div2(f2 a, f2 b) {
    bool f0, f1, g0, g1;
    f2 ds, ns, y;
    ds.x = __builtin_amdgcn_div_scalef(a.x, b.x, false, &g0);
    ds.y = __builtin_amdgcn_div_scalef(a.y, b.y, false, &g1);
    ns.x = __builtin_amdgcn_div_scalef(a.x, b.x, true, &f0);
    ns.y = __builtin_amdgcn_div_scalef(a.y, b.y, true, &f1);
    y.x = __builtin_amdgcn_rcpf(ds.x);
    y.y = __builtin_amdgcn_rcpf(ds.y);
    const f2 one = { 1.0f, 1.0f };
    const f2 nd = -ds;
    const f2 e = __builtin_elementwise_fma(nd, y, one);
    const f2 y1 = __builtin_elementwise_fma(e, y, y);
    const f2 q0 = ns * y1;
    const f2 r0 = __builtin_elementwise_fma(nd, q0, ns);
    const f2 q1 = __builtin_elementwise_fma(r0, y1, q0);
    const f2 r1 = __builtin_elementwise_fma(nd, q1, ns);
    f2 q;
    q.x = __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(r1.x, y1.x, q1.x, f0), b.x, a.x);
    q.y = __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(r1.y, y1.y, q1.y, f1), b.y, a.y);
    return q;
}

template<bool PACKED>
__global__ void __launch_bounds__(256) step_kernel(const float* __restrict__ in, float* __restrict__ out, int iters, float delta, float lam, float dt) {
    constexpr int E = 8;
    float L[E], fs[E], dw[E];
    const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * E;
    for (int j = 0; j < E; ++j) { L[j] = in[base + j]; fs[j] = 0.f; dw[j] = in[base + j] - 0.7f; }
    for (int it = 0; it < iters; ++it) {
        if constexpr (!PACKED) {
            #pragma unroll
            for (int j = 0; j < E; ++j) {
                const float t = (lam * delta) / (1.0f + L[j] * delta);
                fs[j] = fs[j] + t;
                const float drift = fs[j] * lam;
                L[j] = (L[j] + drift * dt) + dw[j] * lam;
            }
        } else {
            #pragma unroll
            for (int j = 0; j < E; j += 2) {
                f2 l = { L[j], L[j + 1] }, s = { fs[j], fs[j + 1] }, w = { dw[j], dw[j + 1] };
                const f2 num = { lam * delta, lam * delta };
                const f2 den = l * delta + 1.0f;
                const f2 t = div2(num, den);
                s = s + t;
                const f2 drift = s * lam;
                l = (l + drift * dt) + w * lam;
                L[j] = l.x; L[j + 1] = l.y; fs[j] = s.x; fs[j + 1] = s.y;
            }
        }
    }
    float acc = 0.f;
    for (int j = 0; j < E; ++j) acc += L[j] + fs[j];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ void check_kernel(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, size_t n, unsigned long long* diff, uint32_t* first) {
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2; i + 1 < n; i += (size_t)gridDim.x * 512) {
        const f2 x = { __uint_as_float(a[i]), __uint_as_float(a[i + 1]) }, y = { __uint_as_float(b[i]), __uint_as_float(b[i + 1]) };
        const f2 q = div2(x, y);
        const float w0 = x.x / y.x, w1 = x.y / y.y;
        const bool s0 = __float_as_uint(w0) == __float_as_uint(q.x) || (w0 != w0 && q.x != q.x);
        const bool s1 = __float_as_uint(w1) == __float_as_uint(q.y) || (w1 != w1 && q.y != q.y);
        if (!s0 || !s1) { atomicAdd(diff, 1ull); atomicMin(first, (uint32_t)i); }
    }
}

int main() {
    const int blocks = 4096, iters = 400;
    std::vector<float> h((size_t)blocks * 256 * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.01f + (float)((i * 2654435761u) % 1000) * 1e-5f;
    float *in, *out; CK(hipMalloc(&in, h.size() * 4)); CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> r0((size_t)blocks * 256), r1((size_t)blocks * 256);
    for (int v = 0; v < 2; ++v) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            if (v == 0) step_kernel<false><<<blocks, 256>>>(in, out, iters, 0.5f, 0.01f, 0.5f); else step_kernel<true><<<blocks, 256>>>(in, out, iters, 0.5f, 0.01f, 0.5f);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("%s: %.3f ms  = %.3f ns per component-step and element on one SIMD\n", v ? "packed (manual division, v_pk_*)" : "scalar (compiler's division)   ", ms,
                                 ms * 1e6 / ((double)blocks * 256 * 8 * iters / 1024.0));
        }
        CK(hipMemcpy((v ? r1 : r0).data(), out, r0.size() * 4, hipMemcpyDeviceToHost));
    }
    printf("results identical: %s\n", memcmp(r0.data(), r1.data(), r0.size() * 4) == 0 ? "yes" : "NO");
    // bit-equality of the division on random bit patterns + specials
    const size_t n = 1u << 26;
    std::vector<uint32_t> a(n), b(n);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
    const uint32_t specials[] = { 0u, 0x80000000u, 0x7f800000u, 0xff800000u, 0x7fc00000u, 1u, 0x007fffffu, 0x00800000u, 0x7f7fffffu, 0x3f800000u, 0x3f7fffffu, 0x3f800001u, 0x00400000u, 0x80000001u };
    for (size_t i = 0; i < n; ++i) {
        uint32_t x = rnd(), y = rnd();
        const uint32_t mode = rnd() & 15u;
        if (mode == 0) x = specials[rnd() % 14];
        if (mode == 1) y = specials[rnd() % 14];
        if (mode == 2) { x = (x & 0x807fffffu) | ((rnd() % 40) << 23); }                       // tiny numerators
        if (mode == 3) { y = (y & 0x807fffffu) | ((215 + rnd() % 40) << 23); }                 // huge denominators
        if (mode == 4) { y = (x & 0x7f800000u) | (y & 0x807fffffu); }                          // quotients near 1
        if (mode == 5) { x = (x & 0x807fffffu) | (127u << 23); y = (y & 0x807fffffu) | (127u << 23); }
        a[i] = x; b[i] = y;
    }
    uint32_t *da, *db; unsigned long long* diff; uint32_t* first;
    CK(hipMalloc(&da, n * 4)); CK(hipMalloc(&db, n * 4)); CK(hipMalloc(&diff, 8)); CK(hipMalloc(&first, 4));
    CK(hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemset(diff, 0, 8)); CK(hipMemset(first, 0xff, 4));
    check_kernel<<<2048, 256>>>(da, db, n, diff, first); CK(hipDeviceSynchronize());
    unsigned long long d; uint32_t f; CK(hipMemcpy(&d, diff, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost));
    printf("manual packed division vs the compiler's a/b on %zu pairs (random bit patterns, specials, denormals): %llu differences (first at %u)\n", n, d, f);
    return 0;
}
