/*
 * rv_float.c — CPU restatement of the reference's fp32 CPU twin (TEST INFRASTRUCTURE, see fm_oracle.h).
 *
 * Follows /root/reference/src/main/java/net/finmath/cuda/cpu/montecarlo/RandomVariableFromFloatArray.java
 * method by method; the line of the Java loop body each case restates is given next to it.
 * Cost model kept on purpose: one single-threaded loop per method call writing a fresh output array
 * (RandomVariableFromFloatArray.java:787-791) — this is what bench.py's cpu_baseline times.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (each Java float operation rounds to fp32 once;
 * a fused multiply-add would change accrue/discount/addProduct, cf. `nvcc -fmad false`, JCudaUtils.java:69-70).
 */
#include "fm_oracle.h"
#include "../include/fmhip.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* java.lang.Math.min/max(float,float): NaN-propagating, -0.0f < +0.0f.  (used at :759, :774; FastMath.min/max
 * at :1167, :1196 have the same contract) */
static inline float jminf(float a, float b) {
    if (a != a) return a;
    if (a == 0.0f && b == 0.0f && signbit(b)) return b;
    return (a <= b) ? a : b;
}
static inline float jmaxf(float a, float b) {
    if (a != a) return a;
    if (a == 0.0f && b == 0.0f && signbit(a)) return b;
    return (a >= b) ? a : b;
}
static inline double jmind(double a, double b) {
    if (a != a) return a;
    if (a == 0.0 && b == 0.0 && signbit(b)) return b;
    return (a <= b) ? a : b;
}
static inline double jmaxd(double a, double b) {
    if (a != a) return a;
    if (a == 0.0 && b == 0.0 && signbit(a)) return b;
    return (a >= b) ? a : b;
}
/* java.lang.Math.pow differs from C99 pow in two special cases: pow(x, NaN) is NaN even for x == 1,
 * and pow(±1, ±inf) is NaN (C99 returns 1 for both). */
static inline double jpow(double x, double y) {
    if (y == 0.0) return 1.0;
    if (y != y) return y;
    if (isinf(y) && fabs(x) == 1.0) return NAN;
    return pow(x, y);
}

#define LOOP(expr) do { for (int64_t i = 0; i < n; i++) { out[i] = (expr); } return 0; } while (0)

int orc_f_v1s0(int opcode, const float* a, int64_t n, float* out) {
    switch (opcode) {
    case FMHIP_OP_SQUARED: LOOP(a[i] * a[i]);                                   /* :875  */
    case FMHIP_OP_SQRT:    LOOP((float)sqrt((double)a[i]));                     /* :890  */
    case FMHIP_OP_EXP:     LOOP((float)exp((double)a[i]));                      /* :905  */
    case FMHIP_OP_LOG:     LOOP((float)log((double)a[i]));                      /* :920  */
    case FMHIP_OP_INVERT:  LOOP(1.0f / a[i]);                                   /* :1296 */
    case FMHIP_OP_ABS:     LOOP(fabsf(a[i]));                                   /* :1311 */
    case FMHIP_OP_SIN:     LOOP((float)sin((double)a[i]));                      /* :935  */
    case FMHIP_OP_COS:     LOOP((float)cos((double)a[i]));                      /* :950  */
    case FMHIP_OP_ISNAN:   LOOP((a[i] != a[i]) ? 1.0f : 0.0f);                  /* :1447 */
    default: return -1;
    }
}

int orc_f_v1s1(int opcode, const float* a, double sd, int64_t n, float* out) {
    const float s = (float)sd;       /* every scalar operand is narrowed first: "(float)value" */
    switch (opcode) {
    case FMHIP_OP_CAP_S:   LOOP(jminf(a[i], s));                                /* :759  */
    case FMHIP_OP_FLOOR_S: LOOP(jmaxf(a[i], s));                                /* :774  */
    case FMHIP_OP_ADD_S:   LOOP(a[i] + s);                                      /* :789  */
    case FMHIP_OP_SUB_S:   LOOP(a[i] - s);                                      /* :804  */
    case FMHIP_OP_BUS_S:   LOOP(-a[i] + s);                                     /* RandomVariableCudaKernel.cu:44-51 */
    case FMHIP_OP_MULT_S:  LOOP(a[i] * s);                                      /* :819  */
    case FMHIP_OP_DIV_S:   LOOP(a[i] / s);                                      /* :834  */
    case FMHIP_OP_VID_S:   LOOP(s / a[i]);                                      /* :1131 (deterministic receiver) */
    case FMHIP_OP_POW_S:   LOOP((float)jpow((double)a[i], (double)s));          /* :849  */
    default: return -1;
    }
}

int orc_f_v2s0(int opcode, const float* a, const float* b, int64_t n, float* out) {
    switch (opcode) {
    case FMHIP_OP_CAP:   LOOP(jminf(a[i], b[i]));                               /* :1167 */
    case FMHIP_OP_FLOOR: LOOP(jmaxf(a[i], b[i]));                               /* :1196 */
    case FMHIP_OP_ADD:   LOOP(a[i] + b[i]);                                     /* :983  */
    case FMHIP_OP_SUB:   LOOP(a[i] - b[i]);                                     /* :1013 */
    case FMHIP_OP_MULT:  LOOP(a[i] * b[i]);                                     /* :1075 */
    case FMHIP_OP_DIV:   LOOP(a[i] / b[i]);                                     /* :1108 */
    default: return -1;
    }
}

int orc_f_v2s1(int opcode, const float* a, const float* b, double sd, int64_t n, float* out) {
    const float s = (float)sd;
    switch (opcode) {
    case FMHIP_OP_ACCRUE:        LOOP(a[i] * (1.0f + b[i] * s));                /* :1224, .cu:224-231 */
    case FMHIP_OP_DISCOUNT:      LOOP(a[i] / (1.0f + b[i] * s));                /* :1252, .cu:234-244 */
    case FMHIP_OP_ADDPRODUCT_VS: LOOP(a[i] + b[i] * s);                         /* :1347, .cu:257-264 */
    default: return -1;
    }
}

int orc_f_v3s0(int opcode, const float* a, const float* b, const float* c, int64_t n, float* out) {
    switch (opcode) {
    case FMHIP_OP_ADDPRODUCT: LOOP(a[i] + b[i] * c[i]);                         /* :1376, .cu:247-254 */
    case FMHIP_OP_ADDRATIO:   LOOP(a[i] + b[i] / c[i]);                         /* :1411 */
    case FMHIP_OP_SUBRATIO:   LOOP(a[i] - b[i] / c[i]);                         /* :1434 */
    case FMHIP_OP_CHOOSE:     LOOP((a[i] >= 0.0) ? b[i] : c[i]);                /* :1281 */
    default: return -1;
    }
}

void orc_f_from_double(const double* in, int64_t n, float* out) {              /* :217-223 */
    for (int64_t i = 0; i < n; i++) out[i] = (float)in[i];
}
void orc_f_to_double(const float* in, int64_t n, double* out) {                /* :225-231 */
    for (int64_t i = 0; i < n; i++) out[i] = in[i];
}

/* ------------------------------------------------------------------ reductions */

double orc_f_average(const float* x, int64_t n) {                              /* :314-334 */
    if (n == 0) return NAN;
    double sum = 0.0, error = 0.0;
    for (int64_t i = 0; i < n; i++) {
        const double value = x[i] - error;
        const double newSum = sum + value;
        error = (newSum - sum) - value;
        sum = newSum;
    }
    return sum / n;
}

double orc_f_average_weighted(const float* x, const float* w, int64_t n) {     /* :337-357 */
    if (n == 0) return NAN;
    double sum = 0.0, error = 0.0;
    for (int64_t i = 0; i < n; i++) {
        const double value = x[i] * (double)w[i] - error;   /* probabilities.get(i) returns double */
        const double newSum = sum + value;
        error = (newSum - sum) - value;
        sum = newSum;
    }
    return sum / n;
}

double orc_f_variance(const float* x, int64_t n) {                             /* :360-382 */
    if (n == 1) return 0.0;
    if (n == 0) return NAN;
    const double average = orc_f_average(x, n);
    double sum = 0.0, errorOfSum = 0.0;
    for (int64_t i = 0; i < n; i++) {
        const double value = (x[i] - average) * (x[i] - average) - errorOfSum;
        const double newSum = sum + value;
        errorOfSum = (newSum - sum) - value;
        sum = newSum;
    }
    return sum / n;
}

double orc_f_variance_weighted(const float* x, const float* w, int64_t n) {    /* :385-407 (no division by n) */
    if (n == 0) return NAN;
    const double average = orc_f_average_weighted(x, w, n);
    double sum = 0.0, errorOfSum = 0.0;
    for (int64_t i = 0; i < n; i++) {
        const double value = (x[i] - average) * (x[i] - average) * (double)w[i] - errorOfSum;
        const double newSum = sum + value;
        errorOfSum = (newSum - sum) - value;
        sum = newSum;
    }
    return sum;
}

double orc_f_min(const float* x, int64_t n) {                                  /* :284-296 */
    double m = 1.7976931348623157e308;
    if (n != 0) m = x[0];
    for (int64_t i = 0; i < n; i++) m = jmind((double)x[i], m);
    return m;
}

double orc_f_max(const float* x, int64_t n) {                                  /* :299-311 */
    double m = -1.7976931348623157e308;
    if (n != 0) m = x[0];
    for (int64_t i = 0; i < n; i++) m = jmaxd((double)x[i], m);
    return m;
}

void orc_f_moments(const float* x, int64_t n, double shift, double out4[4]) {
    double s = 0.0, es = 0.0, q = 0.0, eq = 0.0;
    for (int64_t i = 0; i < n; i++) {
        const double d = x[i] - shift;
        double v = d - es;          double t = s + v;  es = (t - s) - v;  s = t;
        v = d * d - eq;             t = q + v;         eq = (t - q) - v;  q = t;
    }
    out4[0] = s; out4[1] = q; out4[2] = orc_f_min(x, n); out4[3] = orc_f_max(x, n);
}

static int cmp_float(const void* pa, const void* pb) {      /* java.util.Arrays.sort(float[]) total order */
    const float a = *(const float*)pa, b = *(const float*)pb;
    if (a < b) return -1;
    if (a > b) return 1;
    if (a == b) { int sa = signbit(a) != 0, sb = signbit(b) != 0; return sb - sa; } /* -0.0 before 0.0 */
    if (a != a) return (b != b) ? 0 : 1;                    /* NaN last */
    return -1;
}

double orc_f_quantile(const float* x, int64_t n, double quantile) {            /* :473-487 */
    if (n == 0) return NAN;
    float* sorted = (float*)malloc(sizeof(float) * (size_t)n);
    memcpy(sorted, x, sizeof(float) * (size_t)n);
    qsort(sorted, (size_t)n, sizeof(float), cmp_float);
    /* Math.round(double) = floor(x + 0.5) */
    long idx = (long)floor((n + 1) * quantile - 1 + 0.5);
    if (idx < 0) idx = 0;
    if (idx > n - 1) idx = n - 1;
    const double r = sorted[idx];
    free(sorted);
    return r;
}
