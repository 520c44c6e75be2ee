/*
 * fm_oracle.h — CPU oracle for the RandomVariable / BrownianMotion hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product library
 * (libfmhip.so) never links, loads or calls anything declared here.
 *
 * What it restates (file:line relative to /root/reference):
 *   - fp32 element-wise semantics and reductions of the reference's own CPU twin
 *     src/main/java/net/finmath/cuda/cpu/montecarlo/RandomVariableFromFloatArray.java
 *     ("gives exactly the same results" as the GPU class, RandomVariableCuda.java:67-68);
 *   - rounding sequence of accrue/discount/addProduct from RandomVariableCudaKernel.cu:224-264;
 *   - a double-precision variant standing in for finmath-lib's RandomVariableFromDoubleArray
 *     (net.finmath:finmath-lib:5.1.3, pom.xml:29 — NOT vendored under /root/reference);
 *   - java.util.Random (published LCG spec) so the reference test's input stream
 *     `new Random(31415).nextDouble()` (RandomVariableGPUTest.java:194-201) is reproducible without a JVM;
 *   - this repository's own Philox4x32-10 + Box–Muller normal generator specification
 *     (pinned by the published Random123 known-answer vectors for Philox4x32-10).
 *
 * Parity status: the reference holds NO golden vectors or fixture files and cannot be
 * run here (no JVM).  The oracle is pinned by every known-answer value the reference's own tests state
 * for this path (RandomVariableGPUTest.java:82-188; tests/test_oracle_known_answers.py).  Bit-level
 * parity with finmath-lib's double class is "parity unpinned".
 */
#ifndef FM_ORACLE_H
#define FM_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- float twin: element-wise (opcode numbering = include/fmhip.h). Any input may alias none of the outputs. */
int orc_f_v1s0(int opcode, const float* a, int64_t n, float* out);
int orc_f_v1s1(int opcode, const float* a, double s, int64_t n, float* out);
int orc_f_v2s0(int opcode, const float* a, const float* b, int64_t n, float* out);
int orc_f_v2s1(int opcode, const float* a, const float* b, double s, int64_t n, float* out);
int orc_f_v3s0(int opcode, const float* a, const float* b, const float* c, int64_t n, float* out);
/* double[] -> float[] narrowing (RandomVariableFromFloatArray.java:217-223) and back (:225-231) */
void orc_f_from_double(const double* in, int64_t n, float* out);
void orc_f_to_double(const float* in, int64_t n, double* out);

/* ---- float twin: reductions (RandomVariableFromFloatArray.java:284-382) */
double orc_f_average(const float* x, int64_t n);                       /* :314-334 Kahan in double      */
double orc_f_average_weighted(const float* x, const float* w, int64_t n); /* :337-357                    */
double orc_f_variance(const float* x, int64_t n);                      /* :360-382 two-pass Kahan       */
double orc_f_variance_weighted(const float* x, const float* w, int64_t n); /* :385-407                   */
double orc_f_min(const float* x, int64_t n);                           /* :284-296                      */
double orc_f_max(const float* x, int64_t n);                           /* :299-311                      */
/* Σ(x-shift), Σ(x-shift)², Kahan in double — the shape of fmhip_reduce_moments */
void   orc_f_moments(const float* x, int64_t n, double shift, double out4[4]);
double orc_f_quantile(const float* x, int64_t n, double quantile);     /* :473-487                      */

/* ---- double stand-in for RandomVariableFromDoubleArray (same ops, all arithmetic in double) */
int orc_d_v1s0(int opcode, const double* a, int64_t n, double* out);
int orc_d_v1s1(int opcode, const double* a, double s, int64_t n, double* out);
int orc_d_v2s0(int opcode, const double* a, const double* b, int64_t n, double* out);
int orc_d_v2s1(int opcode, const double* a, const double* b, double s, int64_t n, double* out);
int orc_d_v3s0(int opcode, const double* a, const double* b, const double* c, int64_t n, double* out);
double orc_d_average(const double* x, int64_t n);
double orc_d_variance(const double* x, int64_t n);
double orc_d_min(const double* x, int64_t n);
double orc_d_max(const double* x, int64_t n);

/* ---- java.util.Random */
void    orc_java_random_doubles(int64_t seed, int64_t n, double* out);   /* nextDouble() stream */
int32_t orc_java_random_next_int(int64_t seed, int skip);                 /* nextInt() after `skip` calls */

/* ---- Philox4x32-10 and the fmhip normal transform */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* four N(0,1) floats for path block `path_block` (paths 4*pb … 4*pb+3) of stream index (step*F+factor) */
void orc_normal4(int64_t seed, uint64_t path_block, uint32_t stream, float z[4]);
/* out[p] = sqrt_dt * Z(seed, stream, path_offset + p), p in [0,n) */
void orc_bm_increment(int64_t seed, uint32_t stream, int64_t path_offset, int64_t n, float sqrt_dt, float* out);

#ifdef __cplusplus
}
#endif
#endif
