// div_chain_exhaustive.hip — which shortened forms of the in-range division chain (fm_device_math.hpp: div_pair_in_range) return the
// correctly rounded quotient for EVERY pair of operands in [2^-48, 2^48)?  Decided by enumeration, not by argument.
//
// Why enumeration is a proof here.  Inside the range every value of the chain is a normal fp32 number (the reciprocal,
// its residual, the quotient estimates and their exact residuals all lie in (2^-126, 2^127)), v_rcp_f32 maps a normal operand's
// mantissa to the result's mantissa independently of the exponent, and multiplications / fused multiply-adds commute with scaling by
// powers of two as long as nothing leaves the normal range.  The mantissa (and the rounding) of every intermediate therefore depends on
// the MANTISSAS of a and b only: a chain that is right for all 2^23 x 2^23 pairs (a, b) in [1, 2)^2 is right for the whole range
// (signs: every step is odd in a and odd in b).  2^46 quotients: minutes on one MI355X.  As a check of the scaling argument itself, a
// second pass draws exponents at random over the whole range (both edges included) for 2^30 random mantissa pairs.
//
// Reference: the compiler's own `a / b` (-fhip-fp32-correctly-rounded-divide-sqrt: v_div_scale / v_div_fmas / v_div_fixup expansion).
// Chains (y = v_rcp_f32(b); "N" = one Newton step on the reciprocal, "C" = one correction of the quotient by its exact residual):
//   chain 0  N C C   div_pair_in_range as shipped until round 3 (7 packed instructions per pair)
//   chain 1    C C   no Newton step                              (5)
//   chain 2  N C     one correction                              (5)
//   chain 3    C     neither                                     (3: expected to FAIL — shows that the enumeration can tell)
//
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -I finmath-lib-cuda-extensions_amd/csrc benchmarks/div_chain_exhaustive.hip -o div_chain_exhaustive
//   run:   div_chain_exhaustive [first slice = 0] [slices = 128]     (a slice = 2^16 numerator mantissas x all 2^23 denominators)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int CHAINS = 4;

template <int CHAIN>
__device__ __forceinline__ f2 chain(f2 a, f2 b, f2 y) {
    const f2 one = { 1.0f, 1.0f };
    const f2 nb = -b;
    if constexpr (CHAIN == 0 || CHAIN == 2) { const f2 e = __builtin_elementwise_fma(nb, y, one); y = __builtin_elementwise_fma(e, y, y); }
    const f2 q0 = a * y;
    const f2 r0 = __builtin_elementwise_fma(nb, q0, a);
    const f2 q1 = __builtin_elementwise_fma(r0, y, q0);
    if constexpr (CHAIN == 2 || CHAIN == 3) return q1;
    const f2 r1 = __builtin_elementwise_fma(nb, q1, a);
    return __builtin_elementwise_fma(r1, y, q1);
}

struct Counts { unsigned long long bad[CHAINS]; uint32_t first_a[CHAINS], first_b[CHAINS]; };

template <int CHAIN>
__device__ __forceinline__ void check(f2 a, f2 b, f2 y, f2 ref, unsigned& bad, uint32_t& fa, uint32_t& fb) {
    const f2 q = chain<CHAIN>(a, b, y);
    if (__float_as_uint(q.x) != __float_as_uint(ref.x)) { if (!bad) { fa = __float_as_uint(a.x); fb = __float_as_uint(b.x); } ++bad; }
    if (__float_as_uint(q.y) != __float_as_uint(ref.y)) { if (!bad) { fa = __float_as_uint(a.y); fb = __float_as_uint(b.y); } ++bad; }
}

// thread = two denominators (mantissas 2t, 2t + 1: a packed pair); loop = `count` numerator mantissas from a_begin (wave-uniform)
__global__ void __launch_bounds__(256) enumerate(uint32_t a_begin, uint32_t count, Counts* out) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const f2 b = { __uint_as_float(0x3f800000u | (2u * t)), __uint_as_float(0x3f800000u | (2u * t + 1u)) };
    f2 y;
    y.x = __builtin_amdgcn_rcpf(b.x);
    y.y = __builtin_amdgcn_rcpf(b.y);
    unsigned bad[CHAINS] = { 0, 0, 0, 0 };
    uint32_t fa[CHAINS] = { 0, 0, 0, 0 }, fb[CHAINS] = { 0, 0, 0, 0 };
    for (uint32_t i = 0; i < count; ++i) {
        const float av = __uint_as_float(0x3f800000u | (a_begin + i));
        const f2 a = { av, av };
        f2 ref;
        ref.x = a.x / b.x;
        ref.y = a.y / b.y;
        check<0>(a, b, y, ref, bad[0], fa[0], fb[0]);
        check<1>(a, b, y, ref, bad[1], fa[1], fb[1]);
        check<2>(a, b, y, ref, bad[2], fa[2], fb[2]);
        check<3>(a, b, y, ref, bad[3], fa[3], fb[3]);
    }
    for (int c = 0; c < CHAINS; ++c)
        if (bad[c]) { if (atomicAdd(&out->bad[c], (unsigned long long)bad[c]) == 0ull) { out->first_a[c] = fa[c]; out->first_b[c] = fb[c]; } }
}

// the scaling argument, checked: random mantissas AND random exponents over the whole range [2^-48, 2^48), both signs
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ void __launch_bounds__(256) random_exponents(uint32_t seed, uint32_t per_thread, Counts* out) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    unsigned bad[CHAINS] = { 0, 0, 0, 0 };
    uint32_t fa[CHAINS] = { 0, 0, 0, 0 }, fb[CHAINS] = { 0, 0, 0, 0 };
    for (uint32_t i = 0; i < per_thread; ++i) {
        const uint32_t h0 = mix(seed ^ (t * 0x9e3779b9u) ^ (i * 0x85ebca6bu)), h1 = mix(h0 + 0x632be5abu), h2 = mix(h1 ^ 0x12345u), h3 = mix(h2 + 77u);
        auto operand = [](uint32_t m, uint32_t e) {      // biased exponent 79 … 174 = [2^-48, 2^48); every 64th draw an edge of the range
            uint32_t ex = 79u + (e >> 8) % 96u;
            if ((e & 63u) == 0u) ex = (e & 64u) ? 174u : 79u;
            uint32_t mant = m & 0x7fffffu;
            if ((e & 0x3f00u) == 0u) mant = (e & 128u) ? 0x7fffffu : 0u;
            return __uint_as_float((m & 0x80000000u) | (ex << 23) | mant);
        };
        const f2 a = { operand(h0, h2), operand(h1, h3) }, b = { operand(h2 * 3u + h0, h1), operand(h3 * 5u + h1, h0) };
        f2 y, ref;
        y.x = __builtin_amdgcn_rcpf(b.x); y.y = __builtin_amdgcn_rcpf(b.y);
        ref.x = a.x / b.x; ref.y = a.y / b.y;
        check<0>(a, b, y, ref, bad[0], fa[0], fb[0]);
        check<1>(a, b, y, ref, bad[1], fa[1], fb[1]);
        check<2>(a, b, y, ref, bad[2], fa[2], fb[2]);
        check<3>(a, b, y, ref, bad[3], fa[3], fb[3]);
    }
    for (int c = 0; c < CHAINS; ++c)
        if (bad[c]) { if (atomicAdd(&out->bad[c], (unsigned long long)bad[c]) == 0ull) { out->first_a[c] = fa[c]; out->first_b[c] = fb[c]; } }
}

int main(int argc, char** argv) {
    const int first = argc > 1 ? atoi(argv[1]) : 0, slices = argc > 2 ? atoi(argv[2]) : 128;
    if (first < 0 || slices < 1 || first + slices > 128) { printf("slices: 0 … 127\n"); return 2; }
    Counts* dev = nullptr;
    CK(hipMalloc(&dev, sizeof(Counts)));
    CK(hipMemset(dev, 0, sizeof(Counts)));
    const auto t0 = std::chrono::steady_clock::now();
    Counts h{};
    for (int s = first; s < first + slices; ++s) {
        enumerate<<<(1u << 22) / 256u, 256>>>((uint32_t)s << 16, 1u << 16, dev);
        CK(hipDeviceSynchronize());
        if ((s - first) % 8 == 7 || s + 1 == first + slices) {
            CK(hipMemcpy(&h, dev, sizeof h, hipMemcpyDeviceToHost));
            printf("  slices %d … %d done, %.1f s: differences so far %llu / %llu / %llu / %llu\n", first, s, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(),
                   h.bad[0], h.bad[1], h.bad[2], h.bad[3]);
            fflush(stdout);
        }
    }
    const double pairs = (double)slices * 65536.0 * 8388608.0;
    static const char* names[CHAINS] = { "N C C (7 packed instructions per pair, round 3)", "  C C (5: no Newton step on the reciprocal)", "N C   (5: one correction)", "  C   (3: neither)" };
    printf("{\"what\": \"in-range fp32 division chains against the compiler's IEEE a / b on every mantissa pair\", \"numerator_mantissas\": [%u, %u], \"denominator_mantissas\": 8388608, \"quotients\": %.0f, \"seconds\": %.1f, \"chains\": [",
           (unsigned)first << 16, ((unsigned)(first + slices) << 16) - 1u, pairs, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    for (int c = 0; c < CHAINS; ++c)
        printf("%s{\"chain\": \"%s\", \"differences\": %llu, \"first_a_bits\": \"0x%08x\", \"first_b_bits\": \"0x%08x\"}", c ? ", " : "", names[c], h.bad[c], h.first_a[c], h.first_b[c]);
    printf("], ");
    CK(hipMemset(dev, 0, sizeof(Counts)));
    random_exponents<<<4096, 256>>>(0x2545f491u, 1024u, dev);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(&h, dev, sizeof h, hipMemcpyDeviceToHost));
    printf("\"random_exponents\": {\"what\": \"2^31 quotients, mantissas and exponents at random over [2^-48, 2^48), both signs, both edges and all-zero / all-one mantissas over-represented\", \"differences\": [%llu, %llu, %llu, %llu]}}\n",
           h.bad[0], h.bad[1], h.bad[2], h.bad[3]);
    return 0;
}
