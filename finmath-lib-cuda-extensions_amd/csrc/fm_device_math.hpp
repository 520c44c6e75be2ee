// fm_device_math.hpp — per-element arithmetic of every opcode, gfx950 device code.
//
// Contract (DESIGN.md §"Arithmetic contract"): results are bit-identical to the reference's CPU twin
// RandomVariableFromFloatArray.java for + - * / min max abs sqrt choose accrue discount addProduct
// (each elementary operation rounds to fp32 once: this file is compiled with -ffp-contract=off, the
// counterpart of the reference's `nvcc -fmad false`, JCudaUtils.java:69-70), and exp/log/pow/sin/cos are
// evaluated in fp64 and narrowed — the twin computes `(float)Math.exp(realizations[i])` (:905) — so they
// differ from the twin only where two fp64 libms disagree in the last fp64 ulp AND that ulp straddles an
// fp32 rounding boundary (probability ≈ 2^-28 per element).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/fmhip.h"

namespace fm {

// java.lang.Math.min/max(float,float): NaN-propagating, -0.0f < +0.0f (RandomVariableFromFloatArray.java:759,774).
// The reference's CUDA kernels use `a < b ? a : b` (RandomVariableCudaKernel.cu:2-21), which differs from
// its own CPU twin for NaN and signed zeros; the twin (and finmath-lib's double class) is followed here.
__device__ __forceinline__ float jmin(float a, float b) {
    if (a != a) return a;
    if (a == 0.0f && b == 0.0f && (__float_as_uint(b) >> 31)) return b;
    return (a <= b) ? a : b;
}
__device__ __forceinline__ float jmax(float a, float b) {
    if (a != a) return a;
    if (a == 0.0f && b == 0.0f && (__float_as_uint(a) >> 31)) return b;
    return (a >= b) ? a : b;
}
// java.lang.Math.pow special cases that differ from C99 pow (see oracle/rv_float.c jpow).
__device__ __forceinline__ double jpow(double x, double y) {
    if (y == 0.0) return 1.0;
    if (y != y) return y;
    if (isinf(y) && fabs(x) == 1.0) return __builtin_nan("");
    return pow(x, y);
}

// fp64 exp for an fp32 argument, narrowed once.  `(float)Math.exp(realizations[i])`, twin :905.
__device__ __forceinline__ float exp_f(float a) { return (float)exp((double)a); }
__device__ __forceinline__ float log_f(float a) { return (float)log((double)a); }
// The rarely used, register-hungry fp64 functions are kept out of line so that they do not set the VGPR
// budget (and with it the occupancy) of the whole interpreter kernel.
__device__ __noinline__ float sin_f(float a) { return (float)sin((double)a); }
__device__ __noinline__ float cos_f(float a) { return (float)cos((double)a); }
__device__ __noinline__ float pow_f(float a, float s) { return (float)jpow((double)a, (double)s); }
// (float)Math.sqrt((double)a) == correctly rounded fp32 sqrt (double rounding is innocuous for sqrt, 53 >= 2*24+2).
// __builtin_sqrtf lowers to the IEEE-correct expansion (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt);
// HIP's __fsqrt_rn maps to the NATIVE v_sqrt_f32 (≈1 ulp) and must not be used here (measured: 15 % of results off by 1 ulp).
__device__ __forceinline__ float sqrt_f(float a) { return __builtin_sqrtf(a); }

// One element of one opcode, opcode known at compile time (the interpreter switches once per
// instruction on the wave-uniform opcode and then evaluates all of a thread's elements with eval<CODE>).
template <int CODE>
__device__ __forceinline__ float eval(float a, float b, float c, float s) {
    if constexpr (CODE == FMHIP_OP_CAP_S)         return jmin(a, s);
    else if constexpr (CODE == FMHIP_OP_FLOOR_S)  return jmax(a, s);
    else if constexpr (CODE == FMHIP_OP_ADD_S)    return a + s;
    else if constexpr (CODE == FMHIP_OP_SUB_S)    return a - s;
    else if constexpr (CODE == FMHIP_OP_BUS_S)    return -a + s;
    else if constexpr (CODE == FMHIP_OP_MULT_S)   return a * s;
    else if constexpr (CODE == FMHIP_OP_DIV_S)    return a / s;
    else if constexpr (CODE == FMHIP_OP_VID_S)    return s / a;
    else if constexpr (CODE == FMHIP_OP_POW_S)    return pow_f(a, s);
    else if constexpr (CODE == FMHIP_OP_SQUARED)  return a * a;
    else if constexpr (CODE == FMHIP_OP_SQRT)     return sqrt_f(a);
    else if constexpr (CODE == FMHIP_OP_EXP)      return exp_f(a);
    else if constexpr (CODE == FMHIP_OP_LOG)      return log_f(a);
    else if constexpr (CODE == FMHIP_OP_INVERT)   return 1.0f / a;
    else if constexpr (CODE == FMHIP_OP_ABS)      return __uint_as_float(__float_as_uint(a) & 0x7fffffffu);
    else if constexpr (CODE == FMHIP_OP_SIN)      return sin_f(a);
    else if constexpr (CODE == FMHIP_OP_COS)      return cos_f(a);
    else if constexpr (CODE == FMHIP_OP_ISNAN)    return (a != a) ? 1.0f : 0.0f;
    else if constexpr (CODE == FMHIP_OP_CAP)      return jmin(a, b);
    else if constexpr (CODE == FMHIP_OP_FLOOR)    return jmax(a, b);
    else if constexpr (CODE == FMHIP_OP_ADD)      return a + b;
    else if constexpr (CODE == FMHIP_OP_SUB)      return a - b;
    else if constexpr (CODE == FMHIP_OP_MULT)     return a * b;
    else if constexpr (CODE == FMHIP_OP_DIV)      return a / b;
    else if constexpr (CODE == FMHIP_OP_ACCRUE)   { float p = b * s; float d = 1.0f + p; return a * d; }   // .cu:224-231
    else if constexpr (CODE == FMHIP_OP_DISCOUNT) { float p = b * s; float d = 1.0f + p; return a / d; }   // .cu:234-244
    else if constexpr (CODE == FMHIP_OP_ADDPRODUCT_VS) { float p = b * s; return a + p; }                  // .cu:257-264
    else if constexpr (CODE == FMHIP_OP_ADDPRODUCT)    { float p = b * c; return a + p; }                  // .cu:247-254
    else if constexpr (CODE == FMHIP_OP_ADDRATIO) { float q = b / c; return a + q; }                       // twin :1411
    else if constexpr (CODE == FMHIP_OP_SUBRATIO) { float q = b / c; return a - q; }                       // twin :1434
    else if constexpr (CODE == FMHIP_OP_CHOOSE)   return (a >= 0.0f) ? b : c;                              // twin :1281
    else return a;
}

} // namespace fm
