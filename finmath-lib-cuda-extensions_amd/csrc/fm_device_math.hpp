// fm_device_math.hpp — per-element arithmetic of every micro-op, gfx950 device code.
//
// Contract (DESIGN.md §"Arithmetic contract"): results are bit-identical to the reference's CPU twin
// RandomVariableFromFloatArray.java for + - * / min max abs sqrt choose accrue discount addProduct
// (each elementary operation rounds to fp32 once: this file is compiled with -ffp-contract=off, the
// counterpart of the reference's `nvcc -fmad false`, JCudaUtils.java:69-70), and exp/log/pow/sin/cos are
// evaluated in fp64 and narrowed ONCE — the twin computes `(float)Math.exp(realizations[i])` (:905).  The fp64
// intermediates of exp, log, sqrt, sin, cos are accurate enough that the narrowed result equals the twin's for every one
// of the 2^32 fp32 arguments (benchmarks/exhaustive_unary.py, profiles/round01_exhaustive_parity.json), and so does pow
// for each of the 29 exponents tried.
#pragma once
#ifndef __HIPCC_RTC__           // the JIT tier compiles this header with hiprtc, which brings its own runtime declarations
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif
#include "fm_program.h"
#include "fm_log_table.hpp"

namespace fm {

// java.lang.Math.min/max(float,float): NaN-propagating, -0.0f < +0.0f (RandomVariableFromFloatArray.java:759,774).
// The reference's CUDA kernels use `a < b ? a : b` (RandomVariableCudaKernel.cu:2-21), which differs from
// its own CPU twin for NaN and signed zeros; the twin (and finmath-lib's double class) is followed here.
// gfx950 has exactly this operation: v_minimum3_f32 / v_maximum3_f32 are the IEEE 754-2019 minimum / maximum (any NaN
// operand gives NaN, -0 < +0) — ONE instruction.  (The older v_min_f32 / v_max_f32 return the other operand when one is
// NaN; with them Java's semantics cost an unordered compare and a select on top.)  Java leaves the NaN payload
// unspecified, parity treats NaN ≡ NaN.
__device__ __forceinline__ float jmin(float a, float b) { return __builtin_elementwise_minimum(a, b); }
__device__ __forceinline__ float jmax(float a, float b) { return __builtin_elementwise_maximum(a, b); }
// minNum / maxNum flavour (returns the other operand when one is NaN) for the reduction accumulators, whose NaN is read
// off Σx² (fm_kernel_parts.hpp).  (inline asm: the builtin would first canonicalise both inputs with a v_max_f32 x,x each)
__device__ __forceinline__ float hw_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float hw_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// three operands per instruction, same semantics: an accumulator takes two new elements at a time
__device__ __forceinline__ float hw_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float hw_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// java.lang.Math.pow special cases that differ from C99 pow (see oracle/rv_float.c jpow).
__device__ __forceinline__ double jpow(double x, double y) {
    if (y == 0.0) return 1.0;
    if (y != y) return y;
    if (isinf(y) && fabs(x) == 1.0) return __builtin_nan("");
    return pow(x, y);
}

// ---- exp: fp64 evaluation for an fp32 argument, narrowed once (`(float)Math.exp(realizations[i])`, twin :905).
// x = k·ln2 + r, |r| <= ln2/2;  e^r by a degree-10 near-minimax polynomial (2^-48.6);
// 2^k applied with ldexp; the final fp64→fp32 conversion rounds once (RNE, denormals honoured).
// ≈ 20 instructions (16 of them fp64) instead of the ≈ 45 of the generic library exp — exp/log dominate the VALU budget
// of a fused stream, and the kernel has to stay under the HBM roofline.  Verified over all 2^32 inputs: identical to
// `(float)exp((double)x)` of the C library everywhere (benchmarks/exhaustive_unary.py).

// p·r + c with a CONSTANT c held in a scalar register pair.  Written as an explicit three-address v_fma_f64: left to the
// compiler, a Horner step becomes `v_mov_b64 tmp, c; v_fmac_f64 tmp, p, r` (two-address form, c copied first) in most
// places — one extra VALU instruction per step, ≈25 % of a fused exp/log stream — and the 19 coefficients sit in 38 VGPRs.
__device__ __forceinline__ double fma_c(double p, double r, double c) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(p), "v"(r), "s"(c));
    return d;
}

__device__ __forceinline__ float exp_f(float a) {
    // clamp in fp32: ±inf and huge arguments give 0 / +inf anyway, and k stays a small integer.  With the NaN-propagating
    // minimum / maximum a NaN argument runs through the whole evaluation (two instructions; v_med3_f32 drops the NaN and
    // needs a compare and a select afterwards).
    const double x = (double)jmax(jmin(a, 90.0f), -110.0f);
    // k = round(x·log2 e) by the 1.5·2^52 trick: after the addition the integer sits in the low mantissa bits (two's
    // complement), so the int for ldexp is the low dword of t — no v_rndne_f64 / v_cvt_i32_f64.
    const double t = __builtin_fma(x, 1.4426950408889634, 6755399441055744.0);
    const double k = t - 6755399441055744.0;
    const int ki = (int)(uint32_t)__double_as_longlong(t);
    double r = __builtin_fma(k, -6.93147180369123816490e-01, x);   // ln2 hi (low 32 bits zero: k*hi exact)
    r = __builtin_fma(k, -1.90821492927058770002e-10, r);          // ln2 lo
    // e^r = 1 + r + r²/2 + r³·q(r): q = near-minimax polynomial of degree 7 on |r| <= 0.3468 (tools/minimax_coefficients.py;
    // relative error of e^r 2^-48.6 — one Horner step less than the degree-11 Taylor tail, and 3x more accurate)
    double p = 0x1.286f24b3f714bp-22;
    p = fma_c(p, r, 0x1.72ad803971e4dp-19);
    p = fma_c(p, r, 0x1.a019d81272de4p-16);
    p = fma_c(p, r, 0x1.a019c3487562dp-13);
    p = fma_c(p, r, 0x1.6c16c17016625p-10);
    p = fma_c(p, r, 0x1.1111111710d7bp-7);
    p = fma_c(p, r, 0x1.5555555555369p-5);
    p = fma_c(p, r, 0x1.5555555554f90p-3);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return (float)__builtin_ldexp(p, ki);           // NaN: t, k, r, p are NaN, ldexp and the conversion keep it
}

// ---- log: fp64 evaluation for an fp32 argument, narrowed once (twin :920).  Table-driven, no division:
// x = 2^e·m, m in [1/2, 1) (v_frexp_*_f32: denormals honoured);  c = m rounded to 8 mantissa bits (a grid of 257 points,
// c = 1 included);  r = (m - c)/c: the difference is exact in fp32, 1/c comes from the table, |r| <= 2^-9;
// log x = (e·LN2_HI + log_hi(c)) + (e·LN2_LO + log_lo(c) + log1p(r)),  log1p(r) = r + r²·Q(r), Q of degree 3.
// The first bracket is exact (log_hi is a multiple of 2^-40), so the result carries ONE rounding at its own magnitude.
// The centring of the argument on 1 is folded into the table (entries below sqrt(1/2) hold log(2c) - ln2, fm_log_table.hpp):
// x just above 1 (m just above 1/2, e = 1) cancels exactly, x just below 1 has c = 1, log_hi = log_lo = 0, r exact.
// 25 instruction slots (the division-based form this replaces: 38 — frexp, centring, reciprocal seed, Newton step,
// division residual, degree-5 atanh series).  The table sits in LDS (8 KB, log_table_init() at kernel start): one
// ds_read_b128 + one ds_read_b64 per element.
// Zero, negative, infinite and NaN arguments take the hardware v_log_f32 of the mantissa, which has exactly the IEEE
// special values needed (-inf, NaN, +inf, NaN): one class test + one select instead of four compare/select pairs.
// Verified over all 2^32 inputs: identical to `(float)log((double)x)` of the C library everywhere (on the device:
// benchmarks/exhaustive_unary.py; the same operation sequence on the CPU: tools/check_log_table.cpp).
typedef double f64x2 __attribute__((ext_vector_type(2)));
__shared__ double fm_log_lds[FM_LOG_TABLE_ENTRIES * 4];

// Workgroup-wide copy of the table into LDS; every kernel that may evaluate log_f does it once before its first pass.
// In two halves, so that a kernel can put the loads of its first pass between them (one memory round trip instead of two).
constexpr int FM_LOG_TABLE_CHUNKS = (FM_LOG_TABLE_ENTRIES * 2 + FM_BLOCK - 1) / FM_BLOCK;      // 16-byte pieces per thread
__device__ __forceinline__ void log_table_issue(f64x2 (&v)[FM_LOG_TABLE_CHUNKS]) {
    const f64x2* __restrict__ src = reinterpret_cast<const f64x2*>(FM_LOG_TABLE);
#pragma unroll
    for (int k = 0; k < FM_LOG_TABLE_CHUNKS; ++k) {
        const uint32_t i = threadIdx.x + (uint32_t)k * FM_BLOCK;
        v[k] = src[i < (uint32_t)FM_LOG_TABLE_ENTRIES * 2u ? i : 0u];
    }
}
__device__ __forceinline__ void log_table_commit(const f64x2 (&v)[FM_LOG_TABLE_CHUNKS]) {
    f64x2* dst = reinterpret_cast<f64x2*>(fm_log_lds);
#pragma unroll
    for (int k = 0; k < FM_LOG_TABLE_CHUNKS; ++k) {
        const uint32_t i = threadIdx.x + (uint32_t)k * FM_BLOCK;
        if (i < (uint32_t)FM_LOG_TABLE_ENTRIES * 2u) dst[i] = v[k];
    }
    __syncthreads();
}
__device__ __forceinline__ void log_table_init() {
    f64x2 v[FM_LOG_TABLE_CHUNKS];
    log_table_issue(v);
    log_table_commit(v);
}

// SPECIALS: true = the result for zero / negative / infinite / NaN arguments is selected here (v_log_f32 of the mantissa: a
// quarter-rate instruction, 4 issue slots of the 25, for arguments that hardly ever occur); false = the caller handles them
// (log_all: one wave-uniform branch for a group of elements).
template <bool SPECIALS = true>
__device__ __forceinline__ float log_f(float a) {
    const float m32 = __builtin_amdgcn_frexp_mantf(a);             // [0.5, 1), sign of a; ±0, ±inf, NaN pass through
    const int e = __builtin_amdgcn_frexp_expf(a);
    const uint32_t cb = (__float_as_uint(m32) + 0x4000u) & 0xffff8000u;     // nearest grid point: 8 mantissa bits, may be 1.0
    // byte offset of the 32-byte entry = index·32, index = bits 15…23 of cb (0 … 255: exponent of [0.5,1); 256: c = 1.0).
    // Special arguments (discarded below) may index past the table: LDS reads have no side effects.
    // (inline asm: the compiler expands the bit-field extract of a constant field into a shift and a mask)
    uint32_t offset;
    asm("v_bfe_u32 %0, %1, 10, 14" : "=v"(offset) : "v"(cb));     // bits 10…14 of cb are zero
    const char* entry = reinterpret_cast<const char*>(fm_log_lds) + offset;
    const f64x2 t = *reinterpret_cast<const f64x2*>(entry);         // { 1/c, log_hi }
    const double log_lo = *reinterpret_cast<const double*>(entry + 16);
    const double r = (double)(m32 - __uint_as_float(cb)) * t.x;     // fp32 difference exact (both multiples of 2^-24, |·| <= 2^-10)
    const double r2 = r * r;
    double q = FM_LOG1P_Q3;
    q = fma_c(q, r, FM_LOG1P_Q2);
    q = fma_c(q, r, FM_LOG1P_Q1);
    q = fma_c(q, r, FM_LOG1P_Q0);
    const double lp = __builtin_fma(r2, q, r);                      // log1p(r)
    const double ed = (double)e;
    const double hi = __builtin_fma(ed, 6.93147180369123816490e-01, t.y);   // exact
    const double lo = __builtin_fma(ed, 1.90821492927058770002e-10, log_lo);
    const double res = hi + (lo + lp);
    // class mask: sNaN|qNaN|-inf|-normal|-denormal|-0|+0|+inf = everything except +denormal (0x080) and +normal (0x100).
    // For those arguments log2 of the MANTISSA (negative → NaN, ±0 → -inf, +inf → +inf, NaN → NaN) is exactly the IEEE result.
    if constexpr (SPECIALS) {
        const float lg = __builtin_amdgcn_logf(m32);
        return __builtin_amdgcn_classf(a, 0x27f) ? lg : (float)res;
    } else
        return (float)res;
}
__device__ __forceinline__ bool log_special(float a) { return __builtin_amdgcn_classf(a, 0x27f); }
// G elements at once (the specialised kernels): the main path for all of them, ONE wave-uniform test for special arguments
template <int G>
__device__ __forceinline__ void log_all(float* a) {
    float y[G];
    bool special = false;
#pragma unroll
    for (int j = 0; j < G; ++j) { y[j] = log_f<false>(a[j]); special = special || log_special(a[j]); }
    if (__builtin_amdgcn_ballot_w64(special) != 0ull) {     // rare; the volatile asm keeps the compiler from if-converting the block
        asm volatile("; log: zero, negative, infinite or NaN argument in this wave");
#pragma unroll
        for (int j = 0; j < G; ++j) y[j] = log_special(a[j]) ? __builtin_amdgcn_logf(__builtin_amdgcn_frexp_mantf(a[j])) : y[j];
    }
#pragma unroll
    for (int j = 0; j < G; ++j) a[j] = y[j];
}

// ---- FAST math mode (fmhip_set_math_mode(FMHIP_MATH_FAST)): exp and log on the hardware transcendental unit
// (v_exp_f32 / v_log_f32, 1 ulp each) with an fp32 range reduction; results within 2 fp32 ulp of the correctly
// rounded value — the accuracy class of the CUDA expf/logf the reference's own kernels call (RandomVariableCudaKernel.cu
// :119-136 `exp(a[i])`, `log(a[i])` on float operands).  ≈10 and ≈6 VALU instructions instead of 23 and 41.
__device__ __forceinline__ float exp_fast(float a) {
    const float x = __builtin_amdgcn_fmed3f(a, -150.0f, 150.0f);       // keeps k in range; ±inf → ±150 → 0 / +inf below
    const float k = __builtin_rintf(x * 1.44269502f);
    float r = __builtin_fmaf(x, 1.44269502f, -k);                      // fractional part of x·log2(e), product not rounded
    r = __builtin_fmaf(x, 1.92596299e-08f, r);                         // low part of log2(e)
    const float res = __builtin_ldexpf(__builtin_amdgcn_exp2f(r), (int)k);
    return (a != a) ? a : res;
}
__device__ __forceinline__ float log_fast(float a) {
    const bool den = a < 1.17549435e-38f;                               // v_log_f32 does not take denormals: pre-scale by 2^32
    const float as = den ? a * 4294967296.0f : a;
    float l2 = __builtin_amdgcn_logf(as);                               // log2; -inf for 0, NaN for negatives, +inf for +inf
    l2 = den ? l2 - 32.0f : l2;
    return __builtin_fmaf(l2, 0.693147123f, l2 * 5.76999906e-08f);      // ln2 = hi + lo, both positive: ±inf stays ±inf
}

// The rarely used, register-hungry fp64 library functions are kept out of line so that they do not set the VGPR
// budget (and with it the occupancy) of the whole interpreter kernel.
__device__ __noinline__ float sin_f(float a) { return (float)sin((double)a); }
__device__ __noinline__ float cos_f(float a) { return (float)cos((double)a); }
// ---- pow: `(float)Math.pow((double)x, (double)(float)exponent)` (twin :849).
// Fast path for x > 0 finite and a finite exponent: 2^(y·log2 x) with log2 x carried as a double-double — mantissa centred on
// [sqrt(1/2), sqrt(2)), s = (m-1)/(m+1) from an fp32 reciprocal seed, one Newton step and the division residual
// (s = s_hi + s_lo to 2^-88), log m = 2·atanh(s) with a degree-6 minimax tail
// (2^-57.6) and 2/ln2 as a double-double, so that y·log2 x is good to ≈ 2^-54 relative even at |y·log2 x| ≈ 150; then the
// exp polynomial on the fraction.  ≈ 70 instructions instead of the ≈ 300 of the generic library pow; every other
// argument class (x <= 0, ±inf, NaN, exponent ±inf / NaN) takes the library path with Java's special cases (jpow).
__device__ __forceinline__ float pow_pos(float a, float yf) {
    float m32 = __builtin_amdgcn_frexp_mantf(a);
    int e = __builtin_amdgcn_frexp_expf(a);
    const bool lo = m32 < 0.70710678f;
    m32 = lo ? m32 + m32 : m32;
    e = lo ? e - 1 : e;
    const double m = (double)m32, f = m - 1.0, d = m + 1.0;
    const double q0 = (double)__builtin_amdgcn_rcpf(m32 + 1.0f);
    const double q1 = __builtin_fma(__builtin_fma(-d, q0, 1.0), q0, q0);
    const double s_hi = f * q1;
    const double s_lo = __builtin_fma(-s_hi, d, f) * q1;                  // division residual: s = s_hi + s_lo to ≈ 2^-88
    const double z = s_hi * s_hi;
    double g = 0x1.2b5b5fb2eac92p-4;                                      // (atanh(s)/s - 1)/z, tools/minimax_coefficients.py
    g = fma_c(g, z, 0x1.39fe208c33457p-4);
    g = fma_c(g, z, 0x1.7462b69705382p-4);
    g = fma_c(g, z, 0x1.c71c62debf86bp-4);
    g = fma_c(g, z, 0x1.2492492df6947p-3);
    g = fma_c(g, z, 0x1.99999999952b2p-3);
    g = fma_c(g, z, 0x1.5555555555558p-2);
    double at_lo = __builtin_fma(s_hi * z, g, s_lo);                      // atanh(s) = s_hi + at_lo
    at_lo = __builtin_fma(z, s_lo, at_lo);                                // first-order effect of s_lo on the s³ term
    const double c_hi = 0x1.71547652b82fep+1, c_lo = 0x1.777d0ffda0d24p-55;   // 2/ln 2
    const double p_hi = c_hi * s_hi;
    const double p_lo = __builtin_fma(c_hi, s_hi, -p_hi) + __builtin_fma(c_hi, at_lo, c_lo * s_hi);
    const double ed = (double)e;
    const double l_hi = ed + p_hi;                                        // Fast2Sum: |ed| >= 1 > |p_hi|, or ed == 0
    const double l_lo = ((ed - l_hi) + p_hi) + p_lo;
    const double y = (double)yf;
    double P_hi = y * l_hi;
    const double P_lo = __builtin_fma(y, l_hi, -P_hi) + y * l_lo;
    P_hi = __builtin_fmin(__builtin_fmax(P_hi, -1100.0), 1100.0);         // results saturate to 0 / +inf; keeps k an int
    const double t = P_hi + 6755399441055744.0;
    const double k = t - 6755399441055744.0;
    const int ki = (int)(uint32_t)__double_as_longlong(t);
    const double ff = (P_hi - k) + P_lo;
    const double r = __builtin_fma(ff, 0x1.abc9e3b39803fp-56, ff * 0x1.62e42fefa39efp-1);   // ff · ln 2
    // e^r as in exp_f, but with a tail of degree 9 (2^-59.8 instead of 2^-48.6: tools/minimax_coefficients.py): exp_f only
    // has to separate the 2^29 arguments whose result is not trivially 0, 1 or inf; pow sees 2^31 bases per exponent
    double p = 0x1.1f6701e62eb6ap-29;
    p = fma_c(p, r, 0x1.af38e5e7b1a55p-26);
    p = fma_c(p, r, 0x1.27e4e1e60c4eep-22);
    p = fma_c(p, r, 0x1.71de0d950eaadp-19);
    p = fma_c(p, r, 0x1.a01a01a47ebf9p-16);
    p = fma_c(p, r, 0x1.a01a01a7caa3bp-13);
    p = fma_c(p, r, 0x1.6c16c16c167ddp-10);
    p = fma_c(p, r, 0x1.11111111109abp-7);
    p = fma_c(p, r, 0x1.5555555555555p-5);
    p = fma_c(p, r, 0x1.5555555555556p-3);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return (float)__builtin_ldexp(p, ki);
}
__device__ __noinline__ float pow_f(float a, float s) {
    const double x = (double)a;
    // Small integer exponents (wave-uniform test): the product / quotient of exact doubles is rounded ONCE, i.e. it is the
    // correctly rounded power — including the exact ties (x² of a short mantissa) that no approximation can resolve.
    if (s == 2.0f) return (float)(x * x);
    if (s == 3.0f) return (float)((x * x) * x);
    if (s == 4.0f) { const double t = x * x; return (float)(t * t); }
    if (s == -1.0f) return (float)(1.0 / x);
    if (s == -2.0f) return (float)(1.0 / (x * x));
    // half-integer exponents: x^(k+1/2) = x^k·sqrt(x); the exact ties of these powers have an exact square root, so the product is exact
    // (positive bases only: Math.pow(-inf, 1.5) = +inf and Math.pow(-0.0, 2.5) = +0.0 are the library's business)
    if (s == 1.5f && a > 0.0f) return (float)(x * __builtin_sqrt(x));
    if (s == 2.5f && a > 0.0f) return (float)((x * x) * __builtin_sqrt(x));
    // +denormal | +normal base (0x180) with a finite exponent (everything but NaN 0x3 and ±inf 0x204); -normal | -denormal
    // base (0x018) with an INTEGER exponent (wave-uniform test): ±|x|^y on the same path, sign by the parity of y (every
    // float >= 2^24 is an even integer).  The library's power is good to a fraction of an fp64 ulp but not exact, which is
    // what an exact tie needs: (-1.375)^7 = -19487171/2^21 lies exactly between two floats.
    const bool s_finite = __builtin_amdgcn_classf(s, 0x1f8);
    const bool s_integer = s_finite && __builtin_truncf(s) == s;
    const bool negative_base = __builtin_amdgcn_classf(a, 0x018) && s_integer;
    if ((__builtin_amdgcn_classf(a, 0x180) && s_finite) || negative_base) {
        float r = pow_pos(__builtin_fabsf(a), s);
        if (negative_base && __builtin_fabsf(s) < 16777216.0f && (((int)s) & 1) != 0) r = -r;
        if (__builtin_fabsf(r) >= 1.17549435e-38f) return r;
        // Results in the fp32 denormal range keep few bits, so exact ties are common there.  For a small positive integer
        // exponent the square-and-multiply product of doubles is EXACT whenever the power is short enough to be a tie
        // (x = 27·2^-30: x^5 = 14348907·2^-150), and the conversion then rounds it once; everything else: library path.
        if (s_integer && s > 0.0f && s <= 64.0f) {
            double acc = 1.0, base = x;
            for (int n = (int)s; n != 0; n >>= 1) { if (n & 1) acc *= base; base *= base; }
            return (float)acc;
        }
    }
    return (float)jpow(x, (double)s);
}

// (float)Math.sqrt((double)a) == correctly rounded fp32 sqrt (double rounding is innocuous for sqrt, 53 >= 2*24+2).
// Fast path for a in [2^-63, 2^63): hardware reciprocal square root (1 ulp) and ONE Newton step on the exact FMA residual
// (Markstein): r = rsq(a), y0 = a·r, h = r/2, e = a - y0² (one FMA, exact), y = y0 + e·h — 5 instructions instead of the 17 of
// the compiler's IEEE expansion (input scaling for denormals, v_sqrt_f32, two neighbour residuals, selects, class test).
// Verified on the device over ALL 2^32 arguments against that expansion (benchmarks/valu_cost.hip: 0 differences; below
// 2^-100 the residual would underflow, which is why the fast path has a range).  Every other argument — zero, denormal, tiny,
// huge, infinite, NaN, negative — takes __builtin_sqrtf (the IEEE expansion; hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt) behind a WAVE-UNIFORM branch that Monte-Carlo data practically never takes.
// (HIP's __fsqrt_rn maps to the NATIVE v_sqrt_f32, ≈1 ulp, and must not be used on its own: measured 15 % of results off by 1 ulp.)
__device__ __forceinline__ float sqrt_fast_path(float a) {
    const float r = __builtin_amdgcn_rsqf(a);
    const float y0 = a * r, h = 0.5f * r;
    const float e = __builtin_fmaf(-y0, y0, a);
    return __builtin_fmaf(e, h, y0);
}
__device__ __forceinline__ bool sqrt_fast_ok(float a) { return (__float_as_uint(a) - 0x20000000u) < 0x3F000000u; }     // 2^-63 <= a < 2^63
// E elements at once: one range test per element, ONE branch for all of them (the fast sequences interleave freely)
template <int E>
__device__ __forceinline__ void sqrt_all(float (&a)[E]) {
    float y[E];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < E; ++j) { y[j] = sqrt_fast_path(a[j]); ok = ok && sqrt_fast_ok(a[j]); }
    if (__builtin_amdgcn_ballot_w64(!ok) != 0ull) {        // wave-uniform and rare; the volatile asm keeps the compiler from if-converting the block
        asm volatile("; sqrt: IEEE expansion for arguments outside [2^-63, 2^63)");
#pragma unroll
        for (int j = 0; j < E; ++j) y[j] = sqrt_fast_ok(a[j]) ? y[j] : __builtin_sqrtf(a[j]);
    }
#pragma unroll
    for (int j = 0; j < E; ++j) a[j] = y[j];
}
__device__ __forceinline__ float sqrt_f(float a) { float v[1] = { a }; sqrt_all<1>(v); return v[0]; }

// ---- IEEE 754 division, two elements at a time -----------------------------------------------------------------------------
// The correctly rounded fp32 quotient is unique, so every correct implementation is interchangeable bit for bit.  This one is the
// compiler's own expansion of `a / b` (LLVM AMDGPU LowerFDIV32 with denormals on: div_scale x2, rcp, one Newton step on the
// reciprocal, quotient, two residual corrections, div_fmas, div_fixup) with its six multiply-add steps issued as PACKED
// instructions (v_pk_fma_f32 / v_pk_mul_f32: two elements per issue slot).  Left to itself the compiler packs the additions and
// multiplications of neighbouring elements but expands every division in scalar form: 14 issue slots per quotient, 11 this way.
// The LMM drift δλ/(1 + δL) makes the four-step simulation kernels VALU-bound; measured on the model's Euler
// step: 0.426 → 0.376 ns per element and SIMD, 0 differences to `a / b` on 2^26 pairs incl. denormals and specials
// (benchmarks/div_packed.hip).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 div_pair(f32x2 a, f32x2 b) {
    bool n0, n1, d0, d1;
    f32x2 ds, ns, y;
    ds.x = __builtin_amdgcn_div_scalef(a.x, b.x, false, &d0);
    ds.y = __builtin_amdgcn_div_scalef(a.y, b.y, false, &d1);
    ns.x = __builtin_amdgcn_div_scalef(a.x, b.x, true, &n0);
    ns.y = __builtin_amdgcn_div_scalef(a.y, b.y, true, &n1);
    y.x = __builtin_amdgcn_rcpf(ds.x);
    y.y = __builtin_amdgcn_rcpf(ds.y);
    const f32x2 one = { 1.0f, 1.0f };
    const f32x2 nd = -ds;
    const f32x2 e = __builtin_elementwise_fma(nd, y, one);
    const f32x2 y1 = __builtin_elementwise_fma(e, y, y);
    const f32x2 q0 = ns * y1;
    const f32x2 r0 = __builtin_elementwise_fma(nd, q0, ns);
    const f32x2 q1 = __builtin_elementwise_fma(r0, y1, q0);
    const f32x2 r1 = __builtin_elementwise_fma(nd, q1, ns);
    f32x2 q;
    q.x = __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(r1.x, y1.x, q1.x, n0), b.x, a.x);
    q.y = __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(r1.y, y1.y, q1.y, n1), b.y, a.y);
    return q;
}
// ---- the same quotient without the scaling and the fix-up, for operands in a range where both are the identity -----------------
// div_pair spends 8 of its 16 instructions per pair on v_div_scale_f32 (x4), v_div_fmas_f32 (x2) and v_div_fixup_f32 (x2).  By the
// ISA's definition of those three (CDNA3/4 ISA guide, VOP3 opcodes) they do nothing when |a| and |b| both lie in [2^-48, 2^48):
//   v_div_scale_f32 D, VCC, S0, S1 = b, S2 = a returns S0 UNCHANGED with VCC = 0 unless one of its cases applies:
//     a or b zero / NaN / inf        — excluded: both operands are finite, non-zero normal numbers;
//     exponent(a) - exponent(b) >= 96 — the exponents (unbiased) lie in [-48, 47], their difference in [-95, 95];
//     b denormal, 1/b denormal (|b| > 2^126), a/b denormal (|a/b| < 2^-126) — |b| < 2^48 and |a/b| > 2^-96;
//     biased exponent(a) <= 23 (|a| < 2^-103) — |a| >= 2^-48;
//   v_div_fmas_f32 with VCC = 0 is a plain fused multiply-add (the post-scaling by 2^±32 / 2^±64 applies to VCC = 1 only);
//   v_div_fixup_f32 D, S0 = q, S1 = b, S2 = a returns ±|q| with the sign of a/b unless an operand is NaN, infinite or zero, the
//     quotient underflows (exponent(a) - exponent(b) < -150) or b is huge (exponent 255) — none of which can occur; q, the value the
//     multiply-add chain produced, is a non-zero normal number of magnitude in (2^-96, 2^96) and carries the sign of a/b already.
// Inside the range the scaled chain is therefore the same chain on the same numbers — and it is longer than it needs to be.  The
// compiler's expansion corrects the quotient twice (q1 = q0 + r0·y1, q = q1 + r1·y1) because with scaling in play it cannot know how
// good its reciprocal is; inside the range, after ONE Newton step on v_rcp_f32's 1-ulp reciprocal, the FIRST correction already lands on
// the correctly rounded quotient, for every pair of operands: decided by enumeration of all 2^23 x 2^23 mantissa pairs against the
// compiler's own `a / b` (benchmarks/div_chain_exhaustive.hip, 7.04e13 quotients in 40 s on one MI355X, 0 differences:
// profiles/round04_div_chain_exhaustive.json).  Enumeration over mantissas is a proof for the whole range: every value of the chain is
// a normal number there, v_rcp_f32 maps mantissa to mantissa whatever the exponent, and products and fused multiply-adds commute with
// scaling by powers of two (checked as well: 2^31 quotients with exponents drawn over the whole range, both edges included).  Leaving
// out the Newton step instead and keeping both corrections is WRONG for 26 621 mantissa pairs, first (0x3f80d000, 0x3fe45d41).
// 7 instructions per pair (2 v_rcp_f32, 5 packed multiply-adds) + the range test instead of div_pair's 16 (round 3: 9).
__device__ __forceinline__ f32x2 div_pair_in_range(f32x2 a, f32x2 b) {
    f32x2 y;
    y.x = __builtin_amdgcn_rcpf(b.x);
    y.y = __builtin_amdgcn_rcpf(b.y);
    const f32x2 one = { 1.0f, 1.0f };
    const f32x2 nd = -b;
    const f32x2 e = __builtin_elementwise_fma(nd, y, one);
    const f32x2 y1 = __builtin_elementwise_fma(e, y, y);
    const f32x2 q0 = a * y1;
    const f32x2 r0 = __builtin_elementwise_fma(nd, q0, a);
    return __builtin_elementwise_fma(r0, y1, q0);
}
// Range test of a group of operands: all of them lie in [2^-48, 2^48) exactly when the IEEE 754-2019 minimum of their magnitudes is
// >= 2^-48 and the maximum < 2^48 — v_minimum3_f32 / v_maximum3_f32 with |x| source modifiers take in TWO operands per instruction and
// let a NaN through to the comparison, which it fails: one instruction per operand and two comparisons per group (round 3: a key per
// operand by v_lshl_add_u32 plus v_max3_u32 over the keys, 1.5 per operand).  A wave-uniform operand (a scalar of the row block, the
// constant 1) is tested on the scalar unit through its integer key: (bits << 1) drops the sign, the subtraction moves 2^-48 (biased
// exponent 79) to zero, in range exactly when key < FM_DIV_RANGE_SPAN (unsigned).
constexpr uint32_t FM_DIV_RANGE_LOW = 79u << 24, FM_DIV_RANGE_SPAN = 96u << 24;
__device__ __forceinline__ uint32_t div_range_key(float x) { return (__float_as_uint(x) << 1) - FM_DIV_RANGE_LOW; }
struct DivRange {
    float hi = 0.0f, lo = __builtin_huge_valf();
    __device__ __forceinline__ void take(f32x2 v) {
        hi = __builtin_elementwise_maximum(__builtin_elementwise_maximum(hi, __builtin_fabsf(v.x)), __builtin_fabsf(v.y));
        lo = __builtin_elementwise_minimum(__builtin_elementwise_minimum(lo, __builtin_fabsf(v.x)), __builtin_fabsf(v.y));
    }
    __device__ __forceinline__ bool outside() const { return !(lo >= 0x1p-48f) || !(hi < 0x1p48f); }
};

// Numerator and denominator of element pair j of a dividing micro-op (the additions and multiplications in front of the division
// of accrue / discount are part of the micro-op: each rounds once, like the reference's kernels, .cu:234-244)
template <uint32_t CODE, int E>
__device__ __forceinline__ void div_operands(int j, const float (&acc)[E], const float* r1, const float* r2, float s, f32x2& num, f32x2& den) {
    const f32x2 a = { acc[j], acc[j + 1] };
    f32x2 x = { 0.f, 0.f }, y = { 0.f, 0.f };
    if (r1) { x.x = r1[j]; x.y = r1[j + 1]; }
    if (r2) { y.x = r2[j]; y.y = r2[j + 1]; }
    const f32x2 sv = { s, s }, one = { 1.0f, 1.0f };
    if constexpr (CODE == U_INVERT)          { num = one; den = a; }
    else if constexpr (CODE == U_DIV_S)      { num = a; den = sv; }
    else if constexpr (CODE == U_VID_S)      { num = sv; den = a; }
    else if constexpr (CODE == U_DIV)        { num = a; den = x; }
    else if constexpr (CODE == U_VID)        { num = x; den = a; }
    else if constexpr (CODE == U_DISCOUNT_A) { const f32x2 p = x * sv; num = a; den = one + p; }
    else if constexpr (CODE == U_DISCOUNT_B) { const f32x2 p = a * sv; num = x; den = one + p; }
    else                                     { num = x; den = y; }          // addRatio / subRatio
}
// E elements of a dividing micro-op (same results as ueval<CODE> element by element); r1 / r2: operand registers or nullptr.
// The quotients of all E elements by the in-range chain, ONE wave-uniform branch to the full expansion if any operand of any lane
// lies outside the range (Monte-Carlo data: never, in practice) — as sqrt_all and log_all do for their special arguments.
template <uint32_t CODE, int E>
__device__ __forceinline__ void ueval_div_all(float (&out)[E], const float (&acc)[E], const float* r1, const float* r2, float s) {
    static_assert(E % 2 == 0 && fm_uop_divides(CODE), "pairs of elements of a dividing micro-op");
    constexpr bool num_uniform = CODE == U_INVERT || CODE == U_VID_S, den_uniform = CODE == U_DIV_S;
    f32x2 num[E / 2], den[E / 2], q[E / 2];
    DivRange range;
#pragma unroll
    for (int p = 0; p < E / 2; ++p) {
        div_operands<CODE, E>(2 * p, acc, r1, r2, s, num[p], den[p]);
        q[p] = div_pair_in_range(num[p], den[p]);
        if constexpr (!num_uniform) range.take(num[p]);
        if constexpr (!den_uniform) range.take(den[p]);
    }
    bool outside = range.outside();
    if constexpr (num_uniform) outside = outside || div_range_key(num[0].x) >= FM_DIV_RANGE_SPAN;       // scalar unit
    if constexpr (den_uniform) outside = outside || div_range_key(den[0].x) >= FM_DIV_RANGE_SPAN;
    if (__builtin_amdgcn_ballot_w64(outside) != 0ull) {        // wave-uniform and rare; the volatile asm keeps the compiler from if-converting the block
        asm volatile("; division: IEEE expansion with scaling and fix-up for operands outside [2^-48, 2^48)");
#pragma unroll
        for (int p = 0; p < E / 2; ++p) q[p] = div_pair(num[p], den[p]);
    }
#pragma unroll
    for (int p = 0; p < E / 2; ++p) {
        f32x2 r = q[p];
        if constexpr (CODE == U_ADDRATIO_A)      { const f32x2 a = { acc[2 * p], acc[2 * p + 1] }; r = a + q[p]; }
        else if constexpr (CODE == U_SUBRATIO_A) { const f32x2 a = { acc[2 * p], acc[2 * p + 1] }; r = a - q[p]; }
        out[2 * p] = r.x; out[2 * p + 1] = r.y;
    }
}

// discount(a, r, s) = a / (1 + r·s) for SEVERAL numerators over the same (r, s) — the chains of a merged loop kernel (jit.cpp:
// jit_generate_merged_source) all discount by the rate the step has just loaded.  The denominator, its reciprocal, the Newton step and the
// denominator's share of the range test are taken once (div_prepare_discount); each numerator then costs its quotient, the residual, the
// one correction and its own share of the range test (ueval_div_prepared).  The same chain on the same numbers as
// ueval_div_all<U_DISCOUNT_A>: div_pair_in_range(num, den) is a function of its two arguments, the range covers the same operands, and the
// wave-uniform fallback is div_pair(num, den) — bit-identical by construction.
template <int E> struct DivPrepared { f32x2 den[E / 2], nd[E / 2], y1[E / 2]; DivRange range; };
template <int E>
__device__ __forceinline__ void div_prepare_discount(DivPrepared<E>& P, const float (&r1)[E], const float s) {
    static_assert(E % 2 == 0, "pairs of elements");
    const f32x2 sv = { s, s }, one = { 1.0f, 1.0f };
#pragma unroll
    for (int p = 0; p < E / 2; ++p) {
        const f32x2 x = { r1[2 * p], r1[2 * p + 1] };
        const f32x2 pr = x * sv;
        P.den[p] = one + pr;
        f32x2 y;
        y.x = __builtin_amdgcn_rcpf(P.den[p].x);
        y.y = __builtin_amdgcn_rcpf(P.den[p].y);
        P.nd[p] = -P.den[p];
        const f32x2 e = __builtin_elementwise_fma(P.nd[p], y, one);
        P.y1[p] = __builtin_elementwise_fma(e, y, y);
        P.range.take(P.den[p]);
    }
}
template <int E>
__device__ __forceinline__ void ueval_div_prepared(float (&out)[E], const float (&acc)[E], const DivPrepared<E>& P) {
    f32x2 num[E / 2], q[E / 2];
    DivRange range = P.range;
#pragma unroll
    for (int p = 0; p < E / 2; ++p) {
        num[p] = f32x2{ acc[2 * p], acc[2 * p + 1] };
        const f32x2 q0 = num[p] * P.y1[p];
        const f32x2 r0 = __builtin_elementwise_fma(P.nd[p], q0, num[p]);
        q[p] = __builtin_elementwise_fma(r0, P.y1[p], q0);
        range.take(num[p]);
    }
    if (__builtin_amdgcn_ballot_w64(range.outside()) != 0ull) {
        asm volatile("; division: IEEE expansion with scaling and fix-up for operands outside [2^-48, 2^48)");
#pragma unroll
        for (int p = 0; p < E / 2; ++p) q[p] = div_pair(num[p], P.den[p]);
    }
#pragma unroll
    for (int p = 0; p < E / 2; ++p) { out[2 * p] = q[p].x; out[2 * p + 1] = q[p].y; }
}

// pow with a WAVE-UNIFORM exponent (a scalar operand of the row block), all E elements of a lane: the exponents Monte-Carlo code
// actually writes take code of their own, chosen ONCE per micro-op by a scalar comparison — the out-of-line fp64 library path (pow_f:
// ≈ 75 fp64 instructions and a call per element, 2.2-2.35 TB/s for the opcode alone) is left for everything else.  Each special form
// returns (float)Math.pow((double)a, (double)s) bit for bit (all 2^32 arguments per exponent against the oracle:
// profiles/round04_exhaustive_pow.json):
//   s = 2      a·a in fp32: the exact square of a 24-bit number has 48 bits, so the fp64 product is exact and its narrowing IS the fp32 product
//   s = 0.5    the correctly rounded fp32 square root (sqrt_all), except that pow(-0, 0.5) = +0 and pow(-inf, 0.5) = +inf (C99 / Java)
//   s = -1     the correctly rounded fp32 quotient 1 / a (double rounding is innocuous for a quotient: 53 >= 2·24 + 2)
//   s = 3, 4, -2, 1.5, 2.5   the fp64 forms of pow_f, inline (1.5 and 2.5: positive bases; a wave that holds any other base calls the library for those lanes)
template <int E>
__device__ __forceinline__ void pow_all(float (&a)[E], const float s) {
    if (s == 2.0f) {
#pragma unroll
        for (int j = 0; j < E; ++j) a[j] = a[j] * a[j];
    } else if (s == 0.5f) {
        float y[E];
#pragma unroll
        for (int j = 0; j < E; ++j) y[j] = a[j];
        sqrt_all<E>(y);
#pragma unroll
        for (int j = 0; j < E; ++j) a[j] = (a[j] == -__builtin_huge_valf()) ? __builtin_huge_valf() : __builtin_fabsf(y[j]);
    } else if (s == -1.0f) {
        if constexpr (E % 2 == 0) { float q[E]; ueval_div_all<U_INVERT, E>(q, a, nullptr, nullptr, 0.f);
#pragma unroll
            for (int j = 0; j < E; ++j) a[j] = q[j]; }
        else {
#pragma unroll
            for (int j = 0; j < E; ++j) a[j] = 1.0f / a[j]; }
    } else if (s == 3.0f) {
#pragma unroll
        for (int j = 0; j < E; ++j) { const double x = (double)a[j]; a[j] = (float)((x * x) * x); }
    } else if (s == 4.0f) {
#pragma unroll
        for (int j = 0; j < E; ++j) { const double x = (double)a[j], t = x * x; a[j] = (float)(t * t); }
    } else if (s == -2.0f) {
#pragma unroll
        for (int j = 0; j < E; ++j) { const double x = (double)a[j]; a[j] = (float)(1.0 / (x * x)); }
    } else if (s == 1.5f || s == 2.5f) {     // x^k·sqrt(x) for positive bases (pow_f's own form); any other base in the wave: the library's business, lane by lane
        float y[E];
        bool positive = true;
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const double x = (double)a[j], r = __builtin_sqrt(x);
            y[j] = (float)(s == 1.5f ? x * r : (x * x) * r);
            positive = positive && a[j] > 0.0f;
        }
        if (__builtin_amdgcn_ballot_w64(!positive) != 0ull) {
            asm volatile("; pow: a base that is not positive: library path for that lane");
#pragma unroll
            for (int j = 0; j < E; ++j) y[j] = a[j] > 0.0f ? y[j] : pow_f(a[j], s);
        }
#pragma unroll
        for (int j = 0; j < E; ++j) a[j] = y[j];
    } else {
#pragma unroll
        for (int j = 0; j < E; ++j) a[j] = pow_f(a[j], s);
    }
}

// One element of one micro-op (fm_program.h: UOp), micro-op known at compile time: the interpreter switches
// once per instruction on the wave-uniform code and evaluates all of a thread's elements with ueval<CODE>.
//   acc = accumulator, r1/r2 = fetched register operands, s = narrowed scalar.
template <uint32_t CODE>
__device__ __forceinline__ float ueval(float acc, float r1, float r2, float s) {
    if constexpr (CODE == U_LDA)            return r1;
    else if constexpr (CODE == U_SQUARED)   return acc * acc;                                   // twin :875
    else if constexpr (CODE == U_SQRT)      return sqrt_f(acc);                                 // twin :890
    else if constexpr (CODE == U_EXP)       return exp_f(acc);                                  // twin :905
    else if constexpr (CODE == U_LOG)       return log_f<true>(acc);                                  // twin :920
    else if constexpr (CODE == U_EXP_FAST)  return exp_fast(acc);
    else if constexpr (CODE == U_LOG_FAST)  return log_fast(acc);
    else if constexpr (CODE == U_INVERT)    return 1.0f / acc;                                  // twin :1296
    else if constexpr (CODE == U_ABS)       return __uint_as_float(__float_as_uint(acc) & 0x7fffffffu);
    else if constexpr (CODE == U_SIN)       return sin_f(acc);
    else if constexpr (CODE == U_COS)       return cos_f(acc);
    else if constexpr (CODE == U_ISNAN)     return (acc != acc) ? 1.0f : 0.0f;                  // twin :1447
    else if constexpr (CODE == U_CAP_S)     return jmin(acc, s);                                // twin :759
    else if constexpr (CODE == U_FLOOR_S)   return jmax(acc, s);                                // twin :774
    else if constexpr (CODE == U_ADD_S)     return acc + s;
    else if constexpr (CODE == U_SUB_S)     return acc - s;
    else if constexpr (CODE == U_BUS_S)     return -acc + s;                                    // .cu:44-51
    else if constexpr (CODE == U_MULT_S)    return acc * s;
    else if constexpr (CODE == U_DIV_S)     return acc / s;
    else if constexpr (CODE == U_VID_S)     return s / acc;
    else if constexpr (CODE == U_POW_S)     return pow_f(acc, s);                               // twin :849
    else if constexpr (CODE == U_CAP)       return jmin(acc, r1);
    else if constexpr (CODE == U_FLOOR)     return jmax(acc, r1);
    else if constexpr (CODE == U_ADD)       return acc + r1;
    else if constexpr (CODE == U_MULT)      return acc * r1;
    else if constexpr (CODE == U_SUB)       return acc - r1;
    else if constexpr (CODE == U_BUS)       return r1 - acc;
    else if constexpr (CODE == U_DIV)       return acc / r1;
    else if constexpr (CODE == U_VID)       return r1 / acc;
    else if constexpr (CODE == U_ACCRUE_A)  { float p = r1 * s;  float d = 1.0f + p; return acc * d; }   // .cu:224-231
    else if constexpr (CODE == U_ACCRUE_B)  { float p = acc * s; float d = 1.0f + p; return r1 * d; }
    else if constexpr (CODE == U_DISCOUNT_A){ float p = r1 * s;  float d = 1.0f + p; return acc / d; }   // .cu:234-244
    else if constexpr (CODE == U_DISCOUNT_B){ float p = acc * s; float d = 1.0f + p; return r1 / d; }
    else if constexpr (CODE == U_ADDPRODUCT_VS_A) { float p = r1 * s;  return acc + p; }                  // .cu:257-264
    else if constexpr (CODE == U_ADDPRODUCT_VS_B) { float p = acc * s; return r1 + p; }
    else if constexpr (CODE == U_ADDPRODUCT_A)    { float p = r1 * r2;  return acc + p; }                 // .cu:247-254
    else if constexpr (CODE == U_ADDPRODUCT_B)    { float p = acc * r2; return r1 + p; }
    else if constexpr (CODE == U_ADDRATIO_A)      { float q = r1 / r2; return acc + q; }                  // twin :1411
    else if constexpr (CODE == U_SUBRATIO_A)      { float q = r1 / r2; return acc - q; }                  // twin :1434
    else if constexpr (CODE == U_CHOOSE_T)  return (acc >= 0.0f) ? r1 : r2;                               // twin :1281
    else if constexpr (CODE == U_CHOOSE_P)  return (r1 >= 0.0f) ? acc : r2;
    else if constexpr (CODE == U_CHOOSE_N)  return (r1 >= 0.0f) ? r2 : acc;
    else return acc;
}

} // namespace fm
