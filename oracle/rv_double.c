/*
 * rv_double.c — double-precision stand-in for finmath-lib's RandomVariableFromDoubleArray
 * (TEST INFRASTRUCTURE, see fm_oracle.h).
 *
 * The class the north star names as comparator (RandomVariableFromArrayFactory →
 * RandomVariableFromDoubleArray) lives in net.finmath:finmath-lib:5.1.3 (pom.xml:29 of the reference),
 * which is NOT vendored under /root/reference.  What is restated here is the published contract of the
 * net.finmath.stochastic.RandomVariable interface as the in-tree twin implements it
 * (RandomVariableFromFloatArray.java:750-1451) with every operation carried out in double and scalars
 * NOT narrowed.  Bit-level parity with the real class is "parity unpinned"; the GPU path is compared
 * with this variant only within a stated fp32 tolerance.
 */
#include "fm_oracle.h"
#include "../include/fmhip.h"
#include <math.h>

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
static inline double jpow(double x, double y) {
    if (y == 0.0) return 1.0;
    if (y != y) return y;
    if (isinf(y) && fabs(x) == 1.0) return NAN;
    return pow(x, y);
}

#define LOOP(expr) do { for (int64_t i = 0; i < n; i++) { out[i] = (expr); } return 0; } while (0)

int orc_d_v1s0(int opcode, const double* a, int64_t n, double* out) {
    switch (opcode) {
    case FMHIP_OP_SQUARED: LOOP(a[i] * a[i]);
    case FMHIP_OP_SQRT:    LOOP(sqrt(a[i]));
    case FMHIP_OP_EXP:     LOOP(exp(a[i]));
    case FMHIP_OP_LOG:     LOOP(log(a[i]));
    case FMHIP_OP_INVERT:  LOOP(1.0 / a[i]);
    case FMHIP_OP_ABS:     LOOP(fabs(a[i]));
    case FMHIP_OP_SIN:     LOOP(sin(a[i]));
    case FMHIP_OP_COS:     LOOP(cos(a[i]));
    case FMHIP_OP_ISNAN:   LOOP((a[i] != a[i]) ? 1.0 : 0.0);
    default: return -1;
    }
}
int orc_d_v1s1(int opcode, const double* a, double s, int64_t n, double* out) {
    switch (opcode) {
    case FMHIP_OP_CAP_S:   LOOP(jmind(a[i], s));
    case FMHIP_OP_FLOOR_S: LOOP(jmaxd(a[i], s));
    case FMHIP_OP_ADD_S:   LOOP(a[i] + s);
    case FMHIP_OP_SUB_S:   LOOP(a[i] - s);
    case FMHIP_OP_BUS_S:   LOOP(-a[i] + s);
    case FMHIP_OP_MULT_S:  LOOP(a[i] * s);
    case FMHIP_OP_DIV_S:   LOOP(a[i] / s);
    case FMHIP_OP_VID_S:   LOOP(s / a[i]);
    case FMHIP_OP_POW_S:   LOOP(jpow(a[i], s));
    default: return -1;
    }
}
int orc_d_v2s0(int opcode, const double* a, const double* b, int64_t n, double* out) {
    switch (opcode) {
    case FMHIP_OP_CAP:   LOOP(jmind(a[i], b[i]));
    case FMHIP_OP_FLOOR: LOOP(jmaxd(a[i], b[i]));
    case FMHIP_OP_ADD:   LOOP(a[i] + b[i]);
    case FMHIP_OP_SUB:   LOOP(a[i] - b[i]);
    case FMHIP_OP_MULT:  LOOP(a[i] * b[i]);
    case FMHIP_OP_DIV:   LOOP(a[i] / b[i]);
    default: return -1;
    }
}
int orc_d_v2s1(int opcode, const double* a, const double* b, double s, int64_t n, double* out) {
    switch (opcode) {
    case FMHIP_OP_ACCRUE:        LOOP(a[i] * (1.0 + b[i] * s));
    case FMHIP_OP_DISCOUNT:      LOOP(a[i] / (1.0 + b[i] * s));
    case FMHIP_OP_ADDPRODUCT_VS: LOOP(a[i] + b[i] * s);
    default: return -1;
    }
}
int orc_d_v3s0(int opcode, const double* a, const double* b, const double* c, int64_t n, double* out) {
    switch (opcode) {
    case FMHIP_OP_ADDPRODUCT: LOOP(a[i] + b[i] * c[i]);
    case FMHIP_OP_ADDRATIO:   LOOP(a[i] + b[i] / c[i]);
    case FMHIP_OP_SUBRATIO:   LOOP(a[i] - b[i] / c[i]);
    case FMHIP_OP_CHOOSE:     LOOP((a[i] >= 0.0) ? b[i] : c[i]);
    default: return -1;
    }
}

double orc_d_average(const double* x, int64_t n) {
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
double orc_d_variance(const double* x, int64_t n) {
    if (n == 1) return 0.0;
    if (n == 0) return NAN;
    const double average = orc_d_average(x, n);
    double sum = 0.0, errorOfSum = 0.0;
    for (int64_t i = 0; i < n; i++) {
        const double value = (x[i] - average) * (x[i] - average) - errorOfSum;
        const double newSum = sum + value;
        errorOfSum = (newSum - sum) - value;
        sum = newSum;
    }
    return sum / n;
}
double orc_d_min(const double* x, int64_t n) {
    double m = 1.7976931348623157e308;
    if (n != 0) m = x[0];
    for (int64_t i = 0; i < n; i++) m = jmind(x[i], m);
    return m;
}
double orc_d_max(const double* x, int64_t n) {
    double m = -1.7976931348623157e308;
    if (n != 0) m = x[0];
    for (int64_t i = 0; i < n; i++) m = jmaxd(x[i], m);
    return m;
}
