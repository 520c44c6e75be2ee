// check_log_table.cpp — CPU emulation of the table-driven fp64 log of csrc/fm_device_math.hpp (log_f), operation by
// operation (every device instruction has an exact C counterpart: frexpf, integer masks, fp32 subtraction, fp64 fma),
// compared with `(float)log((double)x)` of the C library over ALL positive finite fp32 arguments.
//   g++ -O2 -fopenmp -ffp-contract=off -o /tmp/check_log_table tools/check_log_table.cpp && /tmp/check_log_table
// (development aid for the table / polynomial generator tools/minimax_coefficients.py; the device result itself is
//  checked on the GPU by benchmarks/exhaustive_unary.py)
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#define __device__ static
#include "../finmath-lib-cuda-extensions_amd/csrc/fm_log_table.hpp"

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static float log_table(float a) {
    int e;
    const float m32 = frexpf(a, &e);                                   // [0.5, 1)
    const uint32_t cb = (f2u(m32) + 0x4000u) & 0xffff8000u;             // nearest grid point (8 mantissa bits), may be 1.0
    const float c = u2f(cb);                                            // (entries below sqrt(1/2) carry the -ln2 of the centring)
    const double* T = &fm::FM_LOG_TABLE[((cb >> 15) & 0x1ffu) * 4];
    const double r = (double)(m32 - c) * T[0];                          // the fp32 difference is exact
    const double r2 = r * r;
    double q = fma(FM_LOG1P_Q3, r, FM_LOG1P_Q2);
    q = fma(q, r, FM_LOG1P_Q1);
    q = fma(q, r, FM_LOG1P_Q0);
    const double lp = fma(r2, q, r);
    const double ed = (double)e;
    const double hi = fma(ed, 6.93147180369123816490e-01, T[1]);       // exact
    const double lo = fma(ed, 1.90821492927058770002e-10, T[2]);
    return (float)(hi + (lo + lp));
}

#ifndef CHECK_STRIDE
#define CHECK_STRIDE 1          // every positive finite argument; a larger odd stride for a quick sample (tests/test_log_table_cpu.py)
#endif

int main(void) {
    long long diffs = 0;
#pragma omp parallel for reduction(+ : diffs) schedule(static, 1 << 20)
    for (int64_t u = 1; u < 0x7f800000ll; u += CHECK_STRIDE) {
        const float x = u2f((uint32_t)u);
        const float ref = (float)log((double)x);
        const float got = log_table(x);
        if (f2u(ref) != f2u(got)) {
            ++diffs;
            if (diffs < 20) printf("x = %a: table %a, libm %a\n", x, got, ref);
        }
    }
    printf("positive finite fp32 arguments (stride %d): %lld differences\n", CHECK_STRIDE, diffs);
    return diffs != 0;
}
