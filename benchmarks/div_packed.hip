#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#include "fm_kernel_parts.hpp"      // -I finmath-lib-cuda-extensions_amd/csrc: the product's own device functions
typedef fm::f32x2 f2;

// VARIANT 0: the compiler's scalar expansion of a / b;  1: fm::div_pair (the same expansion, multiply-adds packed);
// 2: fm::ueval_div_all (round 3: the chain without scaling / fix-up for operands in [2^-48, 2^48), ONE wave-uniform branch to
//    div_pair for the rest)
template<int VARIANT>
__global__ void __launch_bounds__(256) step_kernel(const float* __restrict__ in, float* __restrict__ out, int iters, float delta, float lam, float dt) {
    constexpr int E = 8;
    float L[E], fs[E], dw[E];
    const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * E;
    for (int j = 0; j < E; ++j) { L[j] = in[base + j]; fs[j] = 0.f; dw[j] = in[base + j] - 0.7f; }
    for (int it = 0; it < iters; ++it) {
        if constexpr (VARIANT == 2) {
            float t[E], den[E];
            #pragma unroll
            for (int j = 0; j < E; ++j) den[j] = 1.0f + L[j] * delta;
            fm::ueval_div_all<fm::U_VID_S, E>(t, den, nullptr, nullptr, lam * delta);
            #pragma unroll
            for (int j = 0; j < E; ++j) {
                fs[j] = fs[j] + t[j];
                const float drift = fs[j] * lam;
                L[j] = (L[j] + drift * dt) + dw[j] * lam;
            }
        } else if constexpr (VARIANT == 0) {
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
                const f2 t = fm::div_pair(num, den);
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
        const float w0 = x.x / y.x, w1 = x.y / y.y;
        auto same = [](float w, float q) { return __float_as_uint(w) == __float_as_uint(q) || (w != w && q != q); };
        const f2 q = fm::div_pair(x, y);
        float num[2] = { x.x, x.y }, den[2] = { y.x, y.y }, r[2];
        fm::ueval_div_all<fm::U_DIV, 2>(r, num, den, nullptr, 0.0f);                          // range test + either chain, as the kernels do
        bool ok = same(w0, q.x) && same(w1, q.y) && same(w0, r[0]) && same(w1, r[1]);
        // the in-range chain on its own, wherever the range test would let it run
        const bool in0 = fm::div_range_key(x.x) < fm::FM_DIV_RANGE_SPAN && fm::div_range_key(y.x) < fm::FM_DIV_RANGE_SPAN;
        const bool in1 = fm::div_range_key(x.y) < fm::FM_DIV_RANGE_SPAN && fm::div_range_key(y.y) < fm::FM_DIV_RANGE_SPAN;
        const f2 f = fm::div_pair_in_range(x, y);
        if (in0) { ok = ok && same(w0, f.x); atomicAdd(diff + 1, 1ull); }
        if (in1) { ok = ok && same(w1, f.y); atomicAdd(diff + 1, 1ull); }
        if (!ok) { atomicAdd(diff, 1ull); atomicMin(first, (uint32_t)i); }
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
    std::vector<float> r2((size_t)blocks * 256);
    const char* names[3] = { "scalar (compiler's division)          ", "packed (fm::div_pair, v_pk_*)         ", "ranged (fm::ueval_div_all, no scaling)" };
    for (int v = 0; v < 3; ++v) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            if (v == 0) step_kernel<0><<<blocks, 256>>>(in, out, iters, 0.5f, 0.01f, 0.5f);
            else if (v == 1) step_kernel<1><<<blocks, 256>>>(in, out, iters, 0.5f, 0.01f, 0.5f);
            else step_kernel<2><<<blocks, 256>>>(in, out, iters, 0.5f, 0.01f, 0.5f);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("%s: %.3f ms  = %.3f ns per component-step and element on one SIMD\n", names[v], ms,
                                 ms * 1e6 / ((double)blocks * 256 * 8 * iters / 1024.0));
        }
        CK(hipMemcpy((v == 0 ? r0 : v == 1 ? r1 : r2).data(), out, r0.size() * 4, hipMemcpyDeviceToHost));
    }
    printf("results identical: %s\n", memcmp(r0.data(), r1.data(), r0.size() * 4) == 0 && memcmp(r0.data(), r2.data(), r0.size() * 4) == 0 ? "yes" : "NO");
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
        if (mode >= 6 && mode <= 11) { x = (x & 0x807fffffu) | ((76 + rnd() % 102) << 23); y = (y & 0x807fffffu) | ((76 + rnd() % 102) << 23); }   // around and inside [2^-48, 2^48): both edges of the fast path's range
        if (mode == 12) { x = (x & 0x80000000u) | ((79 + 96 * (rnd() & 1)) << 23) | ((rnd() & 1) ? 0u : 0x007fffffu) ; }                      // exactly on an edge, or one ulp inside the next binade
        a[i] = x; b[i] = y;
    }
    uint32_t *da, *db; unsigned long long* diff; uint32_t* first;
    CK(hipMalloc(&da, n * 4)); CK(hipMalloc(&db, n * 4)); CK(hipMalloc(&diff, 16)); CK(hipMalloc(&first, 4));
    CK(hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemset(diff, 0, 16)); CK(hipMemset(first, 0xff, 4));
    check_kernel<<<2048, 256>>>(da, db, n, diff, first); CK(hipDeviceSynchronize());
    unsigned long long dd[2]; uint32_t f; CK(hipMemcpy(dd, diff, 16, hipMemcpyDeviceToHost)); const unsigned long long d = dd[0]; CK(hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost));
    printf("fm::div_pair, fm::ueval_div_all and (on the %llu quotients whose operands pass the range test) fm::div_pair_in_range vs the compiler's a/b on %zu pairs "
           "(random bit patterns, specials, denormals, both edges of the range): %llu differences (first at %u)\n", dd[1], n, d, f);
    return 0;
}
